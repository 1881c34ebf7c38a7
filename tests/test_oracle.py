"""Pin the CPU oracle against the reference's own fixtures and golden vectors.

Fixtures restated from the reference's tests (data only):
  * 3x3 sub-solver contract      linalg/tests/test_linear_solvers.py:13-23, 63-80
  * 8x8 bordered systems         linalg/schur_complement/tests/test_explicit_schur_complement.py:13-55,
                                 test_mpi_explicit_schur_complement.py:22-115
  * synthetic-KKT known answer   examples/tests/test_examples.py:76-99
Golden vectors: tests/golden/reference_vectors.npz (made by tests/golden/make_golden.py,
which runs the reference's solver files themselves).
"""
import numpy as np
import pytest
from scipy.sparse import coo_matrix
from scipy.sparse.linalg import splu

from oracle.schur_complement import MPISchurComplementLinearSolver, SchurComplementLinearSolver
from oracle.subsolvers import ScipyInterface, SymmetricLDLInterface
from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
from parapint_amd.linalg.comm import SerialComm
from parapint_amd.linalg.results import LinearSolverStatus
from parapint_amd.sparse.block_containers import BlockMatrix, BlockVector, MPIBlockMatrix, MPIBlockVector

KNOWN_ANSWER = 0.3163456780448639   # examples/tests/test_examples.py:86, 99


def base_matrix():
    return coo_matrix(([1, 7, 3, 7, 4, 3, 6], ([0, 0, 0, 1, 1, 2, 2], [0, 1, 2, 0, 1, 0, 2])),
                      shape=(3, 3), dtype=np.double)


@pytest.mark.parametrize('cls', [ScipyInterface, SymmetricLDLInterface])
def test_sub_solver_contract(cls, golden):
    mat = base_matrix()
    zero = mat.copy()
    zero.data.fill(0)
    solver = cls(compute_inertia=True) if cls is ScipyInterface else cls()
    assert solver.do_symbolic_factorization(zero).status == LinearSolverStatus.successful
    assert solver.do_numeric_factorization(mat).status == LinearSolverStatus.successful
    for x_true, key in (([1., 2., 3.], 'sub3_x1'), ([4., 2., 3.], 'sub3_x2')):
        x = solver.do_back_solve(mat * np.array(x_true))
        assert np.allclose(x, x_true)
        assert np.allclose(x, golden[key], rtol=1e-12, atol=1e-12)
    assert tuple(solver.get_inertia()) == tuple(golden['sub3_inertia'])


def build_8x8(symmetric, q11, mpi):
    if symmetric:
        k0 = np.array([[1, 0.5], [0.5, 1]]); k2 = np.array([[1, 1], [1, 3.]])
    else:
        k0 = np.array([[1, 1], [0, 1.]]); k2 = np.array([[1, 0], [1, 1.]])
    ks = [k0, np.eye(2), k2]
    a = [np.array([[0, -1], [0, 0.]]), np.array([[-1, 0], [0, -1.]]), np.array([[0, 0], [-1, 0.]])]
    if mpi:
        A = MPIBlockMatrix(4, 4, np.array([[0, 0, 0, -1]] * 4), SerialComm())
        rhs = MPIBlockVector(4, np.array([0, 0, 0, -1]), SerialComm())
    else:
        A = BlockMatrix(4, 4)
        rhs = BlockVector(4)
    for i in range(3):
        A.set_block(i, i, coo_matrix(ks[i]))
        A.set_block(3, i, coo_matrix(a[i]))
    A.set_block(3, 3, coo_matrix(np.array([[0, 0], [0, q11]], dtype=np.double)))
    for i, v in enumerate(([1, 0], [0, 0], [0, 1], [1, 1])):
        rhs.set_block(i, np.array(v, dtype=np.double))
    return A, rhs


@pytest.mark.parametrize('symmetric', [False, True])
@pytest.mark.parametrize('mpi', [False, True])
def test_bordered_8x8(symmetric, mpi, golden):
    key = 'b8_%s_%s' % ('sym' if symmetric else 'unsym', 'mpi' if mpi else 'ser')
    A, rhs = build_8x8(symmetric, 1.0 if mpi else 0.0, mpi)
    cls = MPISchurComplementLinearSolver if mpi else SchurComplementLinearSolver
    solver = cls({i: ScipyInterface(compute_inertia=True) for i in range(3)}, ScipyInterface(compute_inertia=True))
    full = golden[key + '_full']
    x1 = np.linalg.solve(full, golden[key + '_rhs'])
    solver.do_symbolic_factorization(A)
    solver.do_numeric_factorization(A)
    x2 = solver.do_back_solve(rhs)
    assert np.allclose(x1, x2.flatten())
    assert np.allclose(x2.flatten(), golden[key + '_x'], rtol=1e-13, atol=1e-13)
    eig = np.linalg.eigvals(full)
    inertia = (np.count_nonzero(eig > 0), np.count_nonzero(eig < 0), np.count_nonzero(eig == 0))
    assert solver.get_inertia() == inertia == tuple(golden[key + '_inertia'])
    if mpi:
        S = solver.schur_complement.toarray()
        assert np.allclose(S, golden[key + '_S'], rtol=1e-13, atol=1e-13)
        # second numeric + solve on the same object (test_mpi_...:113-115)
        A, rhs = build_8x8(symmetric, 1.0, True)
        solver.do_numeric_factorization(A)
        assert np.allclose(x1, solver.do_back_solve(rhs).flatten())


@pytest.mark.parametrize('symmetric', [True])
def test_ldl_subsolver_matches_scipy_on_symmetric(symmetric, golden):
    A, rhs = build_8x8(True, 1.0, True)
    solver = MPISchurComplementLinearSolver({i: SymmetricLDLInterface() for i in range(3)}, SymmetricLDLInterface())
    solver.do_symbolic_factorization(A)
    solver.do_numeric_factorization(A)
    x = solver.do_back_solve(rhs)
    assert np.allclose(x.flatten(), golden['b8_sym_mpi_x'], rtol=1e-12, atol=1e-12)
    assert solver.get_inertia() == tuple(golden['b8_sym_mpi_inertia'])


@pytest.mark.parametrize('shape', [(3, 20, 2, 4), (4, 50, 3, 6)])
def test_small_synthetic_against_reference_vectors(shape, golden):
    N, n_q, m, n_t = shape
    key = 'syn_%d_%d_%d_%d' % shape
    model = SyntheticKKT(N, n_q, m, n_t)
    kkt = model.build_kkt(comm=SerialComm())
    rhs = model.build_rhs(comm=SerialComm())
    solver = MPISchurComplementLinearSolver({i: ScipyInterface(compute_inertia=True) for i in range(N)},
                                            ScipyInterface(compute_inertia=True))
    solver.do_symbolic_factorization(kkt)
    solver.do_numeric_factorization(kkt)
    x = solver.do_back_solve(rhs)
    assert np.allclose(solver.schur_complement.toarray(), golden[key + '_S'], rtol=1e-12, atol=1e-12)
    assert np.allclose(x.flatten(), golden[key + '_x'], rtol=1e-10, atol=1e-10)
    assert solver.get_inertia() == tuple(golden[key + '_inertia'])
    assert abs(model.check_result(x) - float(golden[key + '_max_err'][0])) < 1e-9


def test_known_answer_full_space_and_schur(golden):
    model = SyntheticKKT(3, 500, 12, 10)
    assert model.block_dim == 12510 and model.nnz_per_block == 53972
    kkt = model.build_kkt(comm=SerialComm())
    rhs = model.build_rhs(comm=SerialComm())
    # full space (test_examples.py:76-86)
    x = splu(kkt.tocoo().tocsc()).solve(rhs.flatten())
    sol = rhs.copy_structure()
    sol.copyfrom(x)
    assert abs(model.check_result(sol) - KNOWN_ANSWER) < 5e-8
    # parallel Schur class (test_examples.py:88-99)
    solver = MPISchurComplementLinearSolver({i: ScipyInterface() for i in range(3)}, ScipyInterface())
    solver.do_symbolic_factorization(kkt)
    solver.do_numeric_factorization(kkt)
    xs = solver.do_back_solve(rhs)
    assert abs(model.check_result(xs) - KNOWN_ANSWER) < 5e-8
    assert abs(model.check_result(xs) - float(golden['known_answer_psc'][0])) < 1e-10
    assert np.allclose(solver.schur_complement.toarray(), golden['known_answer_S'], rtol=1e-9, atol=1e-9)
    assert np.allclose(xs.get_block(3), golden['known_answer_xc'], rtol=1e-9, atol=1e-9)


def test_value_map_of_the_synthetic_kkt_matches_the_assembled_blocks():
    """f2 parity on the host: gathering every COO entry of K_i and A_i from the interface's arrays through the value
    map reproduces ``build_kkt()`` entry for entry (per-block and per-iteration values, both value patterns)."""
    import numpy as np
    from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
    from parapint_amd.linalg.comm import SerialComm
    model = SyntheticKKT(4, 25, 3, 6)
    nsrc, src, coef = model.value_map()
    N = model.n_blocks
    for it in (None, 2):
        kkt = model.build_kkt(comm=SerialComm(), iteration=it)
        for ndx in range(N):
            kv, bv = model.block_values_from_sources(model.block_sources(ndx, it))
            K, A = kkt.get_block(ndx, ndx).tocoo(), kkt.get_block(N, ndx).tocoo()
            assert src.size == K.nnz + A.nnz and nsrc == model.n_y + model.A.nnz
            assert np.array_equal(kv, K.data) and np.array_equal(bv, A.data)
