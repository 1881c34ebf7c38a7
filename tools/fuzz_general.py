"""Fuzz of the general-LU route (parapint_amd/linalg/general_blocks.py): random block-bordered systems with UNSYMMETRIC
sparse diagonal blocks at wild scales (no dominant diagonal), an unsymmetric corner, ScipyInterface objects as sub-solvers;
the solution against the dense matrix [[K, A^T], [A, Q]].  A system counts as bad if it is handed out with a scaled
residual above 2e-8; a refusal (status singular, or the a-posteriori check raising) is accepted only when one of its
diagonal blocks or the system itself is numerically singular (condition number above 1e12) -- the Schur-complement method
of the reference needs every K_i regular as well.
usage: python tools/fuzz_general.py FIRST COUNT [--gpu]      (--gpu: the product engine on the device, else the test interpreter)"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
from scipy.sparse import coo_matrix

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from parapint_amd.linalg import ScipyInterface                                      # noqa: E402
from parapint_amd.linalg.comm import SerialComm                                     # noqa: E402
from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver  # noqa: E402
from parapint_amd.linalg.results import LinearSolverStatus                          # noqa: E402
from parapint_amd.sparse.block_containers import BlockMatrix, BlockVector           # noqa: E402


def one(seed, make_engine):
    rng = np.random.default_rng(seed)
    nb, n, nc = 2 + seed % 3, 8 + seed % 17, 1 + seed % 5
    A, rhs = BlockMatrix(nb + 1, nb + 1), BlockVector(nb + 1)
    N = nb * n + nc
    full = np.zeros((N, N))
    conds = []
    for i in range(nb):
        K = sp.random(n, n, density=0.3, random_state=int(rng.integers(1 << 30)), format='coo')
        K.data = rng.standard_normal(K.nnz) * 10.0 ** rng.integers(-3, 4, K.nnz)
        K = (K + sp.diags(rng.standard_normal(n) * 10.0 ** rng.integers(-6, 2, n))).tocoo()
        Bd = sp.random(nc, n, density=0.4, random_state=int(rng.integers(1 << 30)), format='coo')
        A.set_block(i, i, K)
        A.set_block(nb, i, Bd)
        full[i * n:(i + 1) * n, i * n:(i + 1) * n] = K.toarray()
        full[nb * n:, i * n:(i + 1) * n] = Bd.toarray()
        full[i * n:(i + 1) * n, nb * n:] = Bd.toarray().T
        rhs.set_block(i, rng.standard_normal(n))
        conds.append(np.linalg.cond(K.toarray()))
    Q = rng.standard_normal((nc, nc))
    A.set_block(nb, nb, coo_matrix(Q))
    full[nb * n:, nb * n:] = Q
    rhs.set_block(nb, rng.standard_normal(nc))
    conds.append(np.linalg.cond(full))
    eng = make_engine()
    s = HipSchurComplementLinearSolver({i: ScipyInterface(engine=eng) for i in range(nb)}, ScipyInterface(engine=eng),
                                       comm=SerialComm(), engine=eng)
    try:
        s.do_symbolic_factorization(A)
        st = s.do_numeric_factorization(A, raise_on_error=False).status
        if st != LinearSolverStatus.successful:
            return None if max(conds) > 1e12 else (seed, 'refused', str(st), max(conds))
        x = s.do_back_solve(rhs).flatten()
    except RuntimeError as e:
        return None if max(conds) > 1e12 else (seed, 'refused', str(e)[:60], max(conds))
    b = rhs.flatten()
    r = np.abs(full @ x - b).max() / (np.abs(full).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())
    return (seed, 'inaccurate', r, max(conds)) if not r <= 2e-8 else None


def main():
    gpu = '--gpu' in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    first, count = int(args[0]), int(args[1])
    if gpu:
        from parapint_amd.linalg.hip_engine import HipEngine as make_engine
    else:
        from hostsim_engine import HostSimEngine as make_engine
    bad, t0 = [], time.time()
    for i, seed in enumerate(range(first, first + count)):
        r = one(seed, make_engine)
        if r is not None:
            bad.append(r)
            print(r, flush=True)
        if (i + 1) % 200 == 0:
            print('done', i + 1, 'bad', len(bad), 'elapsed %.0fs' % (time.time() - t0), flush=True)
    print('TOTAL', count, 'bad', len(bad), 'engine', 'HipEngine' if gpu else 'HostSimEngine')


if __name__ == '__main__':
    main()
