"""Writes tests/golden/refinement_case.npz: a block-bordered KKT system from tools/fuzz_solver.py --hard (seed 316) at the
factorisation where the static pivot sequence -- chosen from the values of the symbolic phase -- solves with a backward
error of 0.48 before refinement: the blocks handed to the symbolic phase, the blocks of that factorisation, the right-hand
side.  The solver class must refine it to <= 1e-8 (tests: case_refinement_fixture).  Data only: inputs of this package's
own random generator; the expected output is the dense solve the test computes.

    python tests/golden/make_refinement_case.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def main():
    import fuzz_solver
    import parapint_amd.linalg.hip_schur_complement as hs
    cap = {'sym': None, 'num': []}
    sym0, num0, bs0 = (hs.HipSchurComplementLinearSolver.do_symbolic_factorization, hs.HipSchurComplementLinearSolver.do_numeric_factorization,
                       hs.HipSchurComplementLinearSolver.do_back_solve)

    def blocks_of(m):
        m = m.to_block_matrix() if getattr(m, 'flat_values', None) is not None else m
        N = m.bshape[0] - 1
        return N, {(i, j): m.get_block(i, j).tocoo() for i in range(N + 1) for j in range(N + 1) if m.get_block(i, j) is not None}

    def sym(self, matrix, *a, **k):
        cap['sym'] = blocks_of(matrix)
        return sym0(self, matrix, *a, **k)

    def num(self, matrix, *a, **k):
        cap['last'] = blocks_of(matrix)
        return num0(self, matrix, *a, **k)

    def bs(self, rhs, *a, **k):
        x = bs0(self, rhs, *a, **k)
        if 'case' not in cap and (self.last_residual_first or 0.0) > 1e-3 and self.solve_repairs == 0 and self.last_residual <= 1e-10:
            N = cap['last'][0]
            cap['case'] = (cap['sym'], cap['last'], [np.asarray(rhs.get_block(i)).copy() for i in range(N + 1)], self.last_residual_first)
        return x
    hs.HipSchurComplementLinearSolver.do_symbolic_factorization = sym
    hs.HipSchurComplementLinearSolver.do_numeric_factorization = num
    hs.HipSchurComplementLinearSolver.do_back_solve = bs
    assert fuzz_solver.one(316, hard=True) is None
    (N, s), (_, f), rhs, rho0 = cap['case']
    out = {'N': N, 'rho_before_refinement': rho0}
    for tag, blk in (('sym', s), ('num', f)):
        for (i, j), b in blk.items():
            out['%s_%d_%d_row' % (tag, i, j)] = b.row.astype(np.int32)
            out['%s_%d_%d_col' % (tag, i, j)] = b.col.astype(np.int32)
            out['%s_%d_%d_val' % (tag, i, j)] = b.data.astype(np.double)
            out['%s_%d_%d_shape' % (tag, i, j)] = np.array(b.shape)
    for i, v in enumerate(rhs):
        out['rhs_%d' % i] = v
    np.savez_compressed(os.path.join(HERE, 'refinement_case.npz'), **out)
    print('written: N =', N, 'rho before refinement', rho0)


if __name__ == '__main__':
    main()
