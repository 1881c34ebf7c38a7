"""Generate golden vectors by running the REFERENCE's own solver files.

Run in the build container only (``python tests/golden/make_golden.py``); the output
``tests/golden/*.npz`` is committed, the reference never travels to the GPU box.

What is executed from /root/reference (loaded by path, unmodified):
    parapint/linalg/results.py
    parapint/linalg/base_linear_solver_interface.py
    parapint/linalg/scipy_interface.py
    parapint/linalg/schur_complement/explicit_schur_complement.py
    parapint/linalg/schur_complement/mpi_explicit_schur_complement.py
Their third-party imports that are absent from this image (pyomo's PyNumero block
containers and timer, mpi4py) are satisfied by this repo's own containers
(parapint_amd.sparse.block_containers) and a size-1 communicator, registered under
the expected module names.  ``import parapint`` itself is *not* executed (its
``__init__`` pulls in Pyomo/ASL model code outside the path).

Inputs come from this repo's generator (SyntheticKKT) and the literal matrices of the
reference's tests; outputs are x, S, inertia and residuals.
"""
import importlib.util
import os
import sys
import types

import numpy as np
from scipy.sparse import coo_matrix

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference'

from parapint_amd.sparse import block_containers as bc  # noqa: E402
from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT  # noqa: E402


class _Comm(object):
    """Size-1 stand-in for MPI.COMM_WORLD (only the calls the two solver files make)."""
    rank = 0
    size = 1

    def Get_rank(self):
        return 0

    def Get_size(self):
        return 1

    def allgather(self, x):
        return [x]

    def allreduce(self, x, op=None):
        return x

    def Allreduce(self, src, dst, op=None):
        dst[...] = src

    def Allgatherv(self, src, dst):
        dst[0][...] = src

    def Barrier(self):
        pass

    def Split(self, color, key):
        return self


class _Timer(object):
    def start(self, name):
        pass

    def stop(self, name):
        pass


def _register(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def load_reference():
    _register('pyomo')
    _register('pyomo.common')
    _register('pyomo.common.timing', HierarchicalTimer=_Timer)
    _register('pyomo.contrib')
    _register('pyomo.contrib.pynumero')
    _register('pyomo.contrib.pynumero.sparse', BlockMatrix=bc.BlockMatrix, BlockVector=bc.BlockVector)
    _register('pyomo.contrib.pynumero.sparse.block_vector', BlockVector=bc.BlockVector)
    _register('pyomo.contrib.pynumero.sparse.block_matrix', BlockMatrix=bc.BlockMatrix)
    _register('pyomo.contrib.pynumero.sparse.mpi_block_matrix', MPIBlockMatrix=bc.MPIBlockMatrix)
    _register('pyomo.contrib.pynumero.sparse.mpi_block_vector', MPIBlockVector=bc.MPIBlockVector)
    mpi = types.SimpleNamespace(COMM_WORLD=_Comm(), Comm=_Comm, MAX='max')
    _register('mpi4py', MPI=mpi)
    # package shells (no __init__ executed) so the relative/absolute imports resolve
    for pkg, path in (('parapint', 'parapint'), ('parapint.linalg', 'parapint/linalg'),
                      ('parapint.linalg.schur_complement', 'parapint/linalg/schur_complement')):
        m = _register(pkg)
        m.__path__ = [os.path.join(REF, path)]

    def load(modname, relpath):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        spec.loader.exec_module(mod)
        return mod

    load('parapint.linalg.results', 'parapint/linalg/results.py')
    load('parapint.linalg.base_linear_solver_interface', 'parapint/linalg/base_linear_solver_interface.py')
    sci = load('parapint.linalg.scipy_interface', 'parapint/linalg/scipy_interface.py')
    ser = load('parapint.linalg.schur_complement.explicit_schur_complement',
               'parapint/linalg/schur_complement/explicit_schur_complement.py')
    par = load('parapint.linalg.schur_complement.mpi_explicit_schur_complement',
               'parapint/linalg/schur_complement/mpi_explicit_schur_complement.py')
    return sci.ScipyInterface, ser.SchurComplementLinearSolver, par.MPISchurComplementLinearSolver


def bordered_8x8(symmetric, q11, mpi):
    """The 8x8 system of the reference's SC tests (test_explicit_schur_complement.py:15-31,
    test_mpi_explicit_schur_complement.py:36-58).  symmetric=True replaces the two
    unsymmetric diagonal blocks by their lower-triangle symmetrisation (quirk Q5)."""
    if symmetric:
        k0 = np.array([[1, 0.5], [0.5, 1]], dtype=np.double)
        k2 = np.array([[1, 1], [1, 3]], dtype=np.double)
    else:
        k0 = np.array([[1, 1], [0, 1]], dtype=np.double)
        k2 = np.array([[1, 0], [1, 1]], dtype=np.double)
    k1 = np.eye(2)
    a = [np.array([[0, -1], [0, 0]], dtype=np.double),
         np.array([[-1, 0], [0, -1]], dtype=np.double),
         np.array([[0, 0], [-1, 0]], dtype=np.double)]
    q = np.array([[0, 0], [0, q11]], dtype=np.double)
    if mpi:
        A = bc.MPIBlockMatrix(4, 4, np.array([[0, 0, 0, -1]] * 4), _Comm())
    else:
        A = bc.BlockMatrix(4, 4)
    for i, k in enumerate((k0, k1, k2)):
        A.set_block(i, i, coo_matrix(k))
        A.set_block(3, i, coo_matrix(a[i]))
    A.set_block(3, 3, coo_matrix(q))
    full = np.zeros((8, 8))
    for i, k in enumerate((k0, k1, k2)):
        full[2 * i:2 * i + 2, 2 * i:2 * i + 2] = k
        full[6:8, 2 * i:2 * i + 2] = a[i]
        full[2 * i:2 * i + 2, 6:8] = a[i].T
    full[6:8, 6:8] = q
    if mpi:
        rhs = bc.MPIBlockVector(4, np.array([0, 0, 0, -1]), _Comm())
    else:
        rhs = bc.BlockVector(4)
    rhs.set_block(0, np.array([1, 0], dtype=np.double))
    rhs.set_block(1, np.array([0, 0], dtype=np.double))
    rhs.set_block(2, np.array([0, 1], dtype=np.double))
    rhs.set_block(3, np.array([1, 1], dtype=np.double))
    return A, rhs, full


def main():
    Scipy, Serial, Parallel = load_reference()
    out = {}

    # 1. sub-solver contract (linalg/tests/test_linear_solvers.py:13-23, 63-80)
    mat = coo_matrix(([1, 7, 3, 7, 4, 3, 6], ([0, 0, 0, 1, 1, 2, 2], [0, 1, 2, 0, 1, 0, 2])),
                     shape=(3, 3), dtype=np.double)
    s = Scipy(compute_inertia=True)
    zero = mat.copy()
    zero.data.fill(0)
    s.do_symbolic_factorization(zero)
    s.do_numeric_factorization(mat)
    out['sub3_x1'] = s.do_back_solve(mat * np.array([1., 2., 3.]))
    out['sub3_x2'] = s.do_back_solve(mat * np.array([4., 2., 3.]))
    out['sub3_inertia'] = np.array(s.get_inertia())

    # 2./3. 8x8 bordered systems, reference serial + parallel classes
    for sym in (False, True):
        for q11, tag, cls, mpi in ((0.0, 'ser', Serial, False), (1.0, 'mpi', Parallel, True)):
            A, rhs, full = bordered_8x8(sym, q11, mpi)
            solver = cls(subproblem_solvers={i: Scipy(compute_inertia=True) for i in range(3)},
                         schur_complement_solver=Scipy(compute_inertia=True))
            flat_rhs = rhs.flatten().copy()
            solver.do_symbolic_factorization(A)
            solver.do_numeric_factorization(A)
            x = solver.do_back_solve(rhs)
            key = 'b8_%s_%s' % ('sym' if sym else 'unsym', tag)
            out[key + '_full'] = full
            out[key + '_rhs'] = flat_rhs
            out[key + '_x'] = x.flatten()
            out[key + '_inertia'] = np.array(solver.get_inertia())
            if mpi:
                sc = solver.schur_complement
                out[key + '_S'] = coo_matrix((sc.data, (sc.row, sc.col)), shape=sc.shape).toarray()

    # 4. small synthetic KKTs through the reference's parallel class
    for (N, n_q, m, n_t) in ((3, 20, 2, 4), (4, 50, 3, 6)):
        model = SyntheticKKT(N, n_q, m, n_t)
        kkt = model.build_kkt(comm=_Comm())
        rhs = model.build_rhs(comm=_Comm())
        solver = Parallel(subproblem_solvers={i: Scipy(compute_inertia=True) for i in range(N)},
                          schur_complement_solver=Scipy(compute_inertia=True))
        solver.do_symbolic_factorization(kkt)
        solver.do_numeric_factorization(kkt)
        x = solver.do_back_solve(rhs)
        key = 'syn_%d_%d_%d_%d' % (N, n_q, m, n_t)
        sc = solver.schur_complement
        out[key + '_S'] = coo_matrix((sc.data, (sc.row, sc.col)), shape=sc.shape).toarray()
        out[key + '_x'] = x.flatten()
        out[key + '_inertia'] = np.array(solver.get_inertia())
        full = (kkt.tocoo()).tocsr()
        out[key + '_resid'] = np.array([np.abs(full * x.flatten() - rhs.flatten()).max()])
        out[key + '_max_err'] = np.array([model.check_result(x)])

    # 5. the reference's known answer, through its own parallel class
    #    (examples/tests/test_examples.py:88-99: n_blocks=3, n_q=500, n_y_multiplier=12)
    model = SyntheticKKT(3, 500, 12, 10)
    kkt = model.build_kkt(comm=_Comm())
    rhs = model.build_rhs(comm=_Comm())
    solver = Parallel(subproblem_solvers={i: Scipy() for i in range(3)}, schur_complement_solver=Scipy())
    solver.do_symbolic_factorization(kkt)
    solver.do_numeric_factorization(kkt)
    x = solver.do_back_solve(rhs)
    out['known_answer_psc'] = np.array([model.check_result(x)])
    sc = solver.schur_complement
    out['known_answer_S'] = coo_matrix((sc.data, (sc.row, sc.col)), shape=sc.shape).toarray()
    out['known_answer_xc'] = np.asarray(x.get_block(3))

    np.savez_compressed(os.path.join(HERE, 'reference_vectors.npz'), **out)
    for k in sorted(out):
        print(k, out[k].shape)
    print('known answer via reference psc class:', repr(float(out['known_answer_psc'][0])))


if __name__ == '__main__':
    main()
