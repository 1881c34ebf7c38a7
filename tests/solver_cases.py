"""Parity cases shared by the CPU (host-interpreter engine) and GPU (HIP engine) suites.

Every case drives HipSchurComplementLinearSolver / HipLDLInterface through the reference's
LinearSolverInterface surface and checks against the oracle, the reference's golden vectors
and dense algebra.  `make_engine` returns a fresh engine (None = product default, the GPU)."""
import numpy as np
import scipy.sparse as sp
from scipy.sparse import coo_matrix
from scipy.sparse.linalg import splu

from oracle.schur_complement import MPISchurComplementLinearSolver as OracleMPISC
from oracle.subsolvers import ScipyInterface as OracleScipy
from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
from parapint_amd.linalg.comm import SerialComm
from parapint_amd.linalg.hip_schur_complement import HipLDLInterface, HipSchurComplementLinearSolver
from parapint_amd.linalg.results import LinearSolverStatus
from parapint_amd.sparse.block_containers import BlockMatrix, BlockVector, MPIBlockMatrix, MPIBlockVector

KNOWN_ANSWER = 0.3163456780448639      # parapint/examples/tests/test_examples.py:86, 99
RESID_TOL = 1e-8                        # BASELINE.json: KKT residual <= 1e-8 vs reference


def new_solver(make_engine, n_blocks=0, comm=None, **kwargs):
    eng = make_engine()
    return HipSchurComplementLinearSolver(subproblem_solvers={i: None for i in range(n_blocks)},
                                          schur_complement_solver=None,
                                          comm=SerialComm() if comm is None else comm, engine=eng, **kwargs)


def scaled_residual(K, x, b):
    K = sp.csr_matrix(K)
    r = np.abs(K @ x - b).max()
    return r / (np.abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())


# ---- 3x3 sub-solver contract (linalg/tests/test_linear_solvers.py:13-23, 63-80) ------------------
def case_sub_solver_contract(make_engine, golden):
    mat = coo_matrix(([1, 7, 3, 7, 4, 3, 6], ([0, 0, 0, 1, 1, 2, 2], [0, 1, 2, 0, 1, 0, 2])),
                     shape=(3, 3), dtype=np.double)
    zero = mat.copy()
    zero.data.fill(0)
    # (the reference runs the same body for its SciPy, MUMPS and MA27 wrappers: test_linear_solvers.py:82-99)
    from parapint_amd.linalg import MumpsInterface, ScipyInterface
    for make in (lambda: MumpsInterface(engine=make_engine()), lambda: ScipyInterface(compute_inertia=True, engine=make_engine()),
                 lambda: HipLDLInterface(engine=make_engine())):
        solver = make()
        assert solver.do_symbolic_factorization(zero).status == LinearSolverStatus.successful
        assert solver.do_numeric_factorization(mat).status == LinearSolverStatus.successful
        for x_true, key in (([1., 2., 3.], 'sub3_x1'), ([4., 2., 3.], 'sub3_x2')):
            x = solver.do_back_solve(mat * np.array(x_true))
            assert np.allclose(x, x_true)
            assert np.allclose(x, golden[key], rtol=1e-10, atol=1e-10)
        assert solver.get_inertia() == tuple(golden['sub3_inertia'])
    # MA27-wrapper semantics: only the lower triangle is read
    low = sp.tril(mat).tocoo()
    solver.do_numeric_factorization(low)
    assert np.allclose(solver.do_back_solve(mat * np.array([1., 2., 3.])), [1., 2., 3.])


# ---- 8x8 bordered system, symmetric variant (quirk Q5) -------------------------------------------
def build_8x8(mpi, q11, upper=False):
    k0 = np.array([[1, 0.5], [0.5, 1]])
    k2 = np.array([[1, 1], [1, 3.]])
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
        if upper:
            A.set_block(i, 3, coo_matrix(a[i].T))
    A.set_block(3, 3, coo_matrix(np.array([[0, 0], [0, q11]], dtype=np.double)))
    for i, v in enumerate(([1, 0], [0, 0], [0, 1], [1, 1])):
        rhs.set_block(i, np.array(v, dtype=np.double))
    return A, rhs


def case_bordered_8x8(make_engine, golden, mpi):
    key = 'b8_sym_%s' % ('mpi' if mpi else 'ser')
    A, rhs = build_8x8(mpi, 1.0 if mpi else 0.0)
    solver = new_solver(make_engine, 3)
    assert solver.do_symbolic_factorization(A).status == LinearSolverStatus.successful
    assert solver.do_numeric_factorization(A).status == LinearSolverStatus.successful
    x = solver.do_back_solve(rhs)
    full = golden[key + '_full']
    x1 = np.linalg.solve(full, golden[key + '_rhs'])
    assert np.allclose(x1, x.flatten())
    assert np.allclose(x.flatten(), golden[key + '_x'], rtol=1e-10, atol=1e-10)
    eig = np.linalg.eigvals(full)
    inertia = (np.count_nonzero(eig > 0), np.count_nonzero(eig < 0), np.count_nonzero(eig == 0))
    assert solver.get_inertia() == inertia == tuple(golden[key + '_inertia'])
    if mpi:
        assert np.allclose(solver.get_schur_complement(), golden[key + '_S'], rtol=1e-12, atol=1e-12)
    # the rhs must not be modified (MPI-class behaviour, quirk Q4) and a second numeric + solve works
    assert np.array_equal(rhs.flatten(), golden[key + '_rhs'])
    solver.do_numeric_factorization(A)
    assert np.allclose(x1, solver.do_back_solve(rhs).flatten())
    # result has the structure of the rhs
    assert type(x) is type(rhs) and x.nblocks == 4


# ---- 8x8 bordered system as the reference's tests state it: unsymmetric diagonal blocks (quirk Q5) -----------------
def build_8x8_original(mpi, q11):
    """linalg/schur_complement/tests/test_explicit_schur_complement.py:15-31, test_mpi_explicit_schur_complement.py: the
    diagonal blocks [[1, 1], [0, 1]] and [[1, 0], [1, 1]] are not symmetric; the reference solves them with SuperLU."""
    A, rhs = build_8x8(mpi, q11)
    A.set_block(0, 0, coo_matrix(np.array([[1, 1], [0, 1.]])))
    A.set_block(2, 2, coo_matrix(np.array([[1, 0], [1, 1.]])))
    return A, rhs


def case_bordered_8x8_original(make_engine, golden, mpi):
    """The reference's own inputs through the ScipyInterface route of this package (general_blocks.py): the solution is
    the reference's (tests/golden: generated by the reference's solver classes), the inertia is not offered."""
    from parapint_amd.linalg import ScipyInterface
    key = 'b8_unsym_%s' % ('mpi' if mpi else 'ser')
    A, rhs = build_8x8_original(mpi, 1.0 if mpi else 0.0)
    full = A.toarray()
    full[:6, 6:] = full[6:, :6].T
    assert np.array_equal(full, golden[key + '_full'])          # (the very matrix the reference's test solves)
    eng = make_engine()
    solver = HipSchurComplementLinearSolver(subproblem_solvers={i: ScipyInterface(compute_inertia=True, engine=eng) for i in range(3)},
                                            schur_complement_solver=ScipyInterface(compute_inertia=True, engine=eng),
                                            comm=SerialComm(), engine=eng)
    assert solver.do_symbolic_factorization(A).status == LinearSolverStatus.successful
    assert solver.do_numeric_factorization(A).status == LinearSolverStatus.successful
    x = solver.do_back_solve(rhs)
    x1 = np.linalg.solve(golden[key + '_full'], golden[key + '_rhs'])
    assert np.allclose(x1, x.flatten())
    assert np.allclose(x.flatten(), golden[key + '_x'], rtol=1e-10, atol=1e-10)
    assert solver.last_multiplier_norm <= 1e-12
    assert np.array_equal(rhs.flatten(), golden[key + '_rhs'])
    if mpi:           # (the reference's S = -sum A K^-1 A^T, read out of the embedded system's coupling block)
        assert np.allclose(solver.get_schur_complement(), golden[key + '_S'], rtol=1e-12, atol=1e-12)
    assert type(x) is type(rhs) and x.nblocks == 4
    try:
        solver.get_inertia()
        raise AssertionError('the inertia of an unsymmetric matrix is not offered')
    except RuntimeError as e:
        assert 'not symmetric' in str(e)
    # a second numeric factorisation + solve on the same object (test_mpi_...:113-115)
    solver.do_numeric_factorization(A)
    assert np.allclose(x1, solver.do_back_solve(rhs).flatten())
    # the same object with the symmetric variant of the system: the symmetric path again, inertia and all
    As, rhs_s = build_8x8(mpi, 1.0 if mpi else 0.0)
    ks = 'b8_sym_%s' % ('mpi' if mpi else 'ser')
    assert solver.do_numeric_factorization(As).status == LinearSolverStatus.successful
    assert np.allclose(solver.do_back_solve(rhs_s).flatten(), golden[ks + '_x'], rtol=1e-10, atol=1e-10)
    assert solver.get_inertia() == tuple(golden[ks + '_inertia'])
    # ... and back
    assert solver.do_numeric_factorization(A).status == LinearSolverStatus.successful
    assert np.allclose(solver.do_back_solve(rhs).flatten(), golden[key + '_x'], rtol=1e-10, atol=1e-10)
    # without ScipyInterface objects nothing is looked at: the lower triangle defines a symmetric matrix (MA27's reading) --
    # for this input a singular one ([[1, 1], [1, 1]] from the last diagonal block)
    plain = new_solver(make_engine, 3)
    plain.do_symbolic_factorization(A)
    assert plain.do_numeric_factorization(A, raise_on_error=False).status == LinearSolverStatus.singular


def case_general_blocks_random(make_engine, seeds=range(6)):
    """Random unsymmetric block-bordered systems (sparse diagonal blocks with an unsymmetric PATTERN, unsymmetric corner)
    against a dense solve of the matrix the reference's classes define: [[K, A^T], [A, Q]]; the single-matrix
    ScipyInterface on an unsymmetric and on a triangular matrix."""
    from parapint_amd.linalg import ScipyInterface
    for seed in seeds:
        rng = np.random.default_rng(100 + seed)
        nb, n, nc = 3 + seed % 3, 12 + 3 * seed, 3 + seed % 4
        A = BlockMatrix(nb + 1, nb + 1)
        rhs = BlockVector(nb + 1)
        N = nb * n + nc
        full = np.zeros((N, N))
        for i in range(nb):
            K = sp.random(n, n, density=0.25, random_state=int(rng.integers(1 << 30)), format='coo')
            K = (K + sp.diags(2.0 + rng.random(n))).tocoo()
            Bd = sp.random(nc, n, density=0.3, random_state=int(rng.integers(1 << 30)), format='coo')
            A.set_block(i, i, K)
            A.set_block(nb, i, Bd)
            full[i * n:(i + 1) * n, i * n:(i + 1) * n] = K.toarray()
            full[nb * n:, i * n:(i + 1) * n] = Bd.toarray()
            full[i * n:(i + 1) * n, nb * n:] = Bd.toarray().T
            rhs.set_block(i, rng.standard_normal(n))
        Q = rng.standard_normal((nc, nc)) + 4.0 * np.eye(nc)
        A.set_block(nb, nb, coo_matrix(Q))
        full[nb * n:, nb * n:] = Q
        rhs.set_block(nb, rng.standard_normal(nc))
        eng = make_engine()
        solver = HipSchurComplementLinearSolver(subproblem_solvers={i: ScipyInterface(engine=eng) for i in range(nb)},
                                                schur_complement_solver=ScipyInterface(engine=eng), comm=SerialComm(), engine=eng)
        assert solver.do_symbolic_factorization(A).status == LinearSolverStatus.successful
        assert solver.do_numeric_factorization(A).status == LinearSolverStatus.successful
        x = solver.do_back_solve(rhs)
        assert scaled_residual(full, x.flatten(), rhs.flatten()) <= 1e-10, (seed, scaled_residual(full, x.flatten(), rhs.flatten()))
        assert solver.last_multiplier_norm <= 1e-8 * max(1.0, np.abs(x.flatten()).max())
    rng = np.random.default_rng(7)
    M = (sp.random(30, 30, density=0.2, random_state=3) + sp.diags(3.0 + rng.random(30))).tocoo()
    one = ScipyInterface(compute_inertia=True, engine=make_engine())
    assert one.do_symbolic_factorization(M).status == LinearSolverStatus.successful
    assert one.do_numeric_factorization(M).status == LinearSolverStatus.successful
    b = rng.standard_normal(30)
    assert scaled_residual(M, one.do_back_solve(b), b) <= 1e-12
    try:
        one.get_inertia()
        raise AssertionError('the inertia of an unsymmetric matrix is not offered')
    except RuntimeError:
        pass
    T = sp.tril(M).tocoo()          # (SuperLU reads what it is given: a triangular matrix is a triangular system)
    assert one.do_numeric_factorization(T).status == LinearSolverStatus.successful
    assert scaled_residual(T, one.do_back_solve(b), b) <= 1e-12
    # triangles that differ in the last bits (assembled separately) are the symmetric path: the inertia stays available
    from parapint_amd.linalg import general_blocks
    Sy = (M + M.T).tocsr()
    noisy = (sp.tril(Sy) + sp.triu(Sy, 1) * (1.0 + 2e-16)).tocoo()
    assert general_blocks.is_symmetric(noisy) and not general_blocks.is_symmetric((sp.tril(Sy) + sp.triu(Sy, 1) * (1.0 + 1e-9)).tocoo())
    assert one.do_numeric_factorization(noisy).status == LinearSolverStatus.successful
    assert one.get_inertia()[2] == 0
    Ssym = (M + M.T).tocoo()        # the symmetric path of the same object: inertia is back
    assert one.do_numeric_factorization(Ssym).status == LinearSolverStatus.successful
    assert scaled_residual(Ssym, one.do_back_solve(b), b) <= 1e-12
    eig = np.linalg.eigvalsh(Ssym.toarray())
    assert one.get_inertia() == (int((eig > 0).sum()), int((eig < 0).sum()), 0)


# ---- small synthetic KKTs against the reference's own output -------------------------------------
def case_small_synthetic(make_engine, golden, shape):
    N, n_q, m, n_t = shape
    key = 'syn_%d_%d_%d_%d' % shape
    model = SyntheticKKT(N, n_q, m, n_t)
    kkt = model.build_kkt(comm=SerialComm())
    rhs = model.build_rhs(comm=SerialComm())
    solver = new_solver(make_engine, N)
    solver.do_symbolic_factorization(kkt)
    solver.do_numeric_factorization(kkt)
    x = solver.do_back_solve(rhs)
    S = solver.get_schur_complement()
    Sg = golden[key + '_S']
    assert np.abs(S - Sg).max() <= 1e-9 * np.abs(Sg).max()
    xg = golden[key + '_x']
    assert np.abs(x.flatten() - xg).max() <= 1e-8 * np.abs(xg).max()
    assert solver.get_inertia() == tuple(golden[key + '_inertia'])
    assert abs(model.check_result(x) - float(golden[key + '_max_err'][0])) < 1e-8
    assert len(solver.plan_stats) == 1          # identical patterns -> one batched group


# ---- the reference's known answer (examples/tests/test_examples.py:76-99) ------------------------
def case_known_answer(make_engine, golden):
    model = SyntheticKKT(3, 500, 12, 10)
    kkt = model.build_kkt(comm=SerialComm())
    rhs = model.build_rhs(comm=SerialComm())
    solver = new_solver(make_engine, 3)
    solver.do_symbolic_factorization(kkt)
    solver.do_numeric_factorization(kkt)
    x = solver.do_back_solve(rhs)
    err = model.check_result(x)
    assert abs(err - KNOWN_ANSWER) < 5e-8                 # assertAlmostEqual, 7 places
    assert abs(err - float(golden['known_answer_psc'][0])) < 1e-9
    Sg = golden['known_answer_S']
    assert np.abs(solver.get_schur_complement() - Sg).max() <= 1e-9 * np.abs(Sg).max()
    assert np.allclose(x.get_block(3), golden['known_answer_xc'], rtol=1e-9, atol=1e-9)
    K = kkt.tocoo()
    assert scaled_residual(K, x.flatten(), rhs.flatten()) <= RESID_TOL
    pos, neg, zero = solver.get_inertia()
    assert zero == 0 and pos + neg == K.shape[0]


# ---- configuration-2-shaped system against the oracle (full-space SuperLU + oracle SC) -----------
def case_against_oracle(make_engine, shape, iteration=None, check_full_space=True):
    N, n_q, m, n_t = shape
    model = SyntheticKKT(N, n_q, m, n_t)
    kkt = model.build_kkt(comm=SerialComm(), iteration=iteration)
    rhs = model.build_rhs(comm=SerialComm())
    solver = new_solver(make_engine, N)
    solver.do_symbolic_factorization(kkt)
    res = solver.do_numeric_factorization(kkt)
    assert res.status == LinearSolverStatus.successful
    x = solver.do_back_solve(rhs)
    K = kkt.tocoo().tocsc()
    b = rhs.flatten()
    assert scaled_residual(K, x.flatten(), b) <= RESID_TOL
    n_y = model.n_y
    pos, neg, zero = solver.get_inertia()
    # inertia of the synthetic KKT: (n_y + n_q) positive per block (+ n_theta), (n_y + n_theta) negative
    assert (pos, neg, zero) == (N * (n_y + model.n_q) + n_t, N * (n_y + n_t), 0)
    if check_full_space:
        x_ref = splu(K).solve(b)
        assert np.abs(x.flatten() - x_ref).max() <= 1e-8 * np.abs(x_ref).max()
    return solver, model


def case_oracle_schur(make_engine, shape):
    """S and x against the oracle's restatement of the reference algorithm (n_c solves per block)."""
    N, n_q, m, n_t = shape
    model = SyntheticKKT(N, n_q, m, n_t)
    kkt = model.build_kkt(comm=SerialComm(), iteration=3)
    rhs = model.build_rhs(comm=SerialComm())
    oracle = OracleMPISC({i: OracleScipy() for i in range(N)}, OracleScipy())
    oracle.do_symbolic_factorization(kkt)
    oracle.do_numeric_factorization(kkt)
    x_o = oracle.do_back_solve(rhs)
    solver = new_solver(make_engine, N)
    solver.do_symbolic_factorization(kkt)
    solver.do_numeric_factorization(kkt)
    x = solver.do_back_solve(rhs)
    S_o = oracle.schur_complement.toarray()
    assert np.abs(solver.get_schur_complement() - S_o).max() <= 1e-9 * np.abs(S_o).max()
    assert np.abs(x.flatten() - x_o.flatten()).max() <= 1e-8 * np.abs(x_o.flatten()).max()


# ---- heterogeneous blocks: several pattern groups, nonzero Q, 2x2 pivots, duplicates (Q7) --------
def random_kkt_block(n_x, n_c, seed, zero_h=0.25):
    """Sparse saddle-point block [[H, J^T], [J, -1e-3 I]] with some zero Hessian diagonals;
    redrawn (seed + 100) until it is comfortably nonsingular."""
    while True:
        rng = np.random.default_rng(seed)
        h = rng.uniform(0.5, 2.0, size=n_x)
        h[rng.random(n_x) < zero_h] = 0.0
        J = sp.random(n_c, n_x, density=0.2, random_state=seed, data_rvs=lambda k: rng.normal(size=k)) + \
            2.0 * sp.eye(n_c, n_x)
        K = sp.bmat([[sp.diags(h), J.T], [J, -1e-3 * sp.eye(n_c)]]).tocoo()
        if np.linalg.cond(K.toarray()) < 1e6:
            return K
        seed += 100


def case_heterogeneous(make_engine):
    rng = np.random.default_rng(11)
    sizes = [(12, 5), (12, 5), (20, 8), (12, 5), (20, 8), (7, 3)]
    seeds = [1, 1, 2, 1, 2, 3]                      # equal seed -> equal pattern -> same group
    nc = 4
    nb = len(sizes)
    A = BlockMatrix(nb + 1, nb + 1)
    rhs = BlockVector(nb + 1)
    for i, ((n_x, n_c), seed) in enumerate(zip(sizes, seeds)):
        K = random_kkt_block(n_x, n_c, seed)
        vals = K.data * rng.uniform(0.8, 1.2, size=K.data.size)
        K = coo_matrix((vals, (K.row, K.col)), shape=K.shape)
        K = (K + K.T).tocoo()                       # symmetric values, both triangles
        if i == 3:                                  # same pattern, shuffled COO order + split duplicates (Q7)
            perm = rng.permutation(K.nnz)
            r, c, d = K.row[perm], K.col[perm], K.data[perm]
            K = coo_matrix((np.concatenate([0.25 * d, 0.75 * d]), (np.concatenate([r, r]), np.concatenate([c, c]))),
                           shape=K.shape)
        n = K.shape[0]
        prng = np.random.default_rng(1000 + seeds[i])            # border pattern fixed per pattern family
        cols = prng.choice(n, size=3, replace=False)
        B = coo_matrix((rng.normal(size=3), (prng.choice(nc, size=3), cols)), shape=(nc, n))
        A.set_block(i, i, K)
        A.set_block(nb, i, B)
        rhs.set_block(i, rng.normal(size=n))
    Q = np.diag([0.5, 0.0, 1.0, 2.0])
    Q[1, 0] = Q[0, 1] = 0.1
    A.set_block(nb, nb, coo_matrix(Q))
    rhs.set_block(nb, rng.normal(size=nc))
    solver = new_solver(make_engine, nb)
    solver.do_symbolic_factorization(A)
    assert solver.do_numeric_factorization(A).status == LinearSolverStatus.successful
    x = solver.do_back_solve(rhs)
    assert len(solver.plan_stats) == 3
    # dense reference of the full symmetric system
    full = np.zeros(A.shape)
    off = np.concatenate([[0], np.cumsum(A.row_block_sizes())])
    for i in range(nb):
        Kd = A.get_block(i, i).toarray()
        full[off[i]:off[i + 1], off[i]:off[i + 1]] = np.tril(Kd) + np.tril(Kd, -1).T
        Bd = A.get_block(nb, i).toarray()
        full[off[nb]:, off[i]:off[i + 1]] = Bd
        full[off[i]:off[i + 1], off[nb]:] = Bd.T
    full[off[nb]:, off[nb]:] = Q
    x_ref = np.linalg.solve(full, rhs.flatten())
    assert np.abs(x.flatten() - x_ref).max() <= 1e-8 * np.abs(x_ref).max()
    ev = np.linalg.eigvalsh(full)
    assert solver.get_inertia() == (int((ev > 0).sum()), int((ev < 0).sum()), 0)
    # values change, pattern and COO order change for one block: numeric again
    K0 = A.get_block(0, 0)
    perm = rng.permutation(K0.nnz)
    A.set_block(0, 0, coo_matrix((1.5 * K0.data[perm], (K0.row[perm], K0.col[perm])), shape=K0.shape))
    full[off[0]:off[1], off[0]:off[1]] *= 1.5
    solver.do_numeric_factorization(A)
    x = solver.do_back_solve(rhs)
    x_ref = np.linalg.solve(full, rhs.flatten())
    assert np.abs(x.flatten() - x_ref).max() <= 1e-8 * np.abs(x_ref).max()


# ---- inertia-correction loop: the regularised KKT has extra diagonal blocks (pattern grows) --------
def case_inertia_correction_pattern_growth(make_engine):
    """interior_point.py:364-392 regularises with `kkt.set_block(.., coef * identity)` on blocks that were empty
    (sc_ip_interface.py:1736-1757) and adds coef*I to the Hessian and to Q: the numeric call then sees a pattern
    that is a superset of the symbolic one.  MUMPS re-does its analysis (mumps_interface.py:82-83); so do we, once,
    on the union pattern; afterwards both patterns factorise without further planning."""
    rng = np.random.default_rng(5)
    n_x, n_c, nc, nb = 10, 4, 3, 5
    A = BlockMatrix(nb + 1, nb + 1)
    rhs = BlockVector(nb + 1)
    Ks, Js, Hs, Bs = [], [], [], []
    for i in range(nb):
        h = rng.uniform(0.5, 2.0, size=n_x)
        h[rng.random(n_x) < 0.3] = 0.0
        hidx = np.flatnonzero(h)                                     # Hessian diagonal only where nonzero
        J = (sp.random(n_c, n_x, density=0.3, random_state=10 + i, data_rvs=lambda k: rng.normal(size=k)) +
             2.0 * sp.eye(n_c, n_x)).tocoo()
        H = coo_matrix((h[hidx], (hidx, hidx)), shape=(n_x, n_x))
        K = sp.bmat([[H, J.T], [J, None]]).tocoo()                  # (2,2) block empty: no diagonal entries there
        B = coo_matrix((rng.normal(size=nc), (np.arange(nc), rng.choice(n_x, nc, replace=False))),
                       shape=(nc, n_x + n_c))
        A.set_block(i, i, K)
        A.set_block(nb, i, B)
        rhs.set_block(i, rng.normal(size=n_x + n_c))
        Ks.append(K); Js.append(J); Hs.append(H); Bs.append(B)
    A.set_block(nb, nb, coo_matrix((nc, nc)))
    rhs.set_block(nb, rng.normal(size=nc))

    def dense(Amat, q):
        off = np.concatenate([[0], np.cumsum([n_x + n_c] * nb), [nb * (n_x + n_c) + nc]])
        full = np.zeros((off[-1], off[-1]))
        for i in range(nb):
            Kd = Amat.get_block(i, i).toarray()
            full[off[i]:off[i + 1], off[i]:off[i + 1]] = np.tril(Kd) + np.tril(Kd, -1).T
            Bd = Amat.get_block(nb, i).toarray()
            full[off[nb]:, off[i]:off[i + 1]] = Bd
            full[off[i]:off[i + 1], off[nb]:] = Bd.T
        full[off[nb]:, off[nb]:] = q
        return full

    def check(Amat, q):
        full = dense(Amat, q)
        ev = np.linalg.eigvalsh(full)
        want = (int((ev > 1e-10).sum()), int((ev < -1e-10).sum()), int((np.abs(ev) <= 1e-10).sum()))
        res = solver.do_numeric_factorization(Amat, raise_on_error=False)
        singular_block = False
        for i in range(nb):                     # the Schur-complement method needs every K_i nonsingular
            Kd = Amat.get_block(i, i).toarray()
            sv = np.linalg.svd(np.tril(Kd) + np.tril(Kd, -1).T, compute_uv=False)
            singular_block = singular_block or sv.min() <= 1e-11 * sv.max()
        if singular_block or want[2] > 0:
            # what the reference's sub-solver reports (ma27_interface.py:126-136) and the inertia loop acts on
            assert res.status == LinearSolverStatus.singular
            return 'singular'
        assert res.status == LinearSolverStatus.successful
        assert solver.get_inertia() == want
        x = solver.do_back_solve(rhs)
        x_ref = np.linalg.solve(full, rhs.flatten())
        assert np.abs(x.flatten() - x_ref).max() <= 1e-8 * np.abs(x_ref).max()
        return 'ok'

    solver = new_solver(make_engine, nb)
    solver.do_symbolic_factorization(A)
    assert check(A, np.zeros((nc, nc))) == 'singular'               # block 0 is singular: the loop regularises
    for coef in (1e-4, 1e-2):                                        # two retries of the inertia loop
        R = BlockMatrix(nb + 1, nb + 1)
        for i in range(nb):
            Hreg = (Hs[i] + coef * sp.identity(n_x, format='coo')).tocoo()            # regularize_hessian
            Creg = (-coef) * sp.identity(n_c, format='coo')                             # regularize_equality_gradient
            R.set_block(i, i, sp.bmat([[Hreg, Js[i].T], [Js[i], Creg]]).tocoo())
            R.set_block(nb, i, Bs[i])
        R.set_block(nb, nb, (coef * sp.identity(nc, format='coo')).tocoo())
        assert check(R, coef * np.eye(nc)) == 'ok'
    # back to an unregularised matrix (the next IP iteration): a subset of the union pattern, no new plan needed
    A2 = BlockMatrix(nb + 1, nb + 1)
    for i in range(nb):
        K = Ks[i].copy()
        K.data = K.data * rng.uniform(0.9, 1.1, size=K.data.size)
        K = ((K + K.T) * 0.5).tocoo()
        A2.set_block(i, i, K)
        A2.set_block(nb, i, Bs[i])
    A2.set_block(nb, nb, coo_matrix((nc, nc)))
    groups_before = [id(g) for g in solver._groups]
    check(A2, np.zeros((nc, nc)))
    assert [id(g) for g in solver._groups] == groups_before        # no re-plan for a subset pattern


# ---- error behaviour ------------------------------------------------------------------------------
def case_pivot_order_refresh(make_engine):
    """The pivot sequence is static per pattern group and fixed from the values the symbolic phase saw.  A later
    matrix with the same pattern can make one of its 1x1 pivots exactly zero although the matrix is nonsingular
    (here: Hessian diagonals that vanish while the constraint rows keep the KKT block regular).  MA27 pivots
    dynamically and factorises such a matrix (ma27_interface.py:124-136 reports singular only for a singular one),
    so the solver must not hand `singular` to the inertia-correction loop: it orders again from the values that
    broke and factorises once more."""
    rng = np.random.default_rng(11)
    n_x, n_c, nc, nb = 8, 3, 2, 6
    Js, Bs = [], []
    for i in range(nb):
        J = (sp.random(n_c, n_x, density=0.4, random_state=30 + i, data_rvs=lambda k: rng.normal(size=k)) +
             2.0 * sp.eye(n_c, n_x)).tocoo()
        B = coo_matrix((rng.normal(size=nc), (np.arange(nc), rng.choice(n_x, nc, replace=False))),
                       shape=(nc, n_x + n_c))
        Js.append(J); Bs.append(B)

    def kkt(hs):
        A = BlockMatrix(nb + 1, nb + 1)
        for i in range(nb):
            H = coo_matrix((hs[i], (np.arange(n_x), np.arange(n_x))), shape=(n_x, n_x))   # explicit zeros stay
            A.set_block(i, i, sp.bmat([[H, Js[i].T], [Js[i], None]]).tocoo())
            A.set_block(nb, i, Bs[i])
        A.set_block(nb, nb, coo_matrix((nc, nc)))
        return A

    def dense(A):
        m = n_x + n_c
        full = np.zeros((nb * m + nc, nb * m + nc))
        for i in range(nb):
            Kd = A.get_block(i, i).toarray()
            full[i * m:(i + 1) * m, i * m:(i + 1) * m] = np.tril(Kd) + np.tril(Kd, -1).T
            Bd = A.get_block(nb, i).toarray()
            full[nb * m:, i * m:(i + 1) * m] = Bd
            full[i * m:(i + 1) * m, nb * m:] = Bd.T
        return full

    rhs = BlockVector(nb + 1)
    for i in range(nb):
        rhs.set_block(i, rng.normal(size=n_x + n_c))
    rhs.set_block(nb, rng.normal(size=nc))

    h0 = [rng.uniform(1.0, 3.0, size=n_x) for _ in range(nb)]       # what the symbolic phase sees: all strong
    h1 = [h.copy() for h in h0]

    def vanish(i, count):
        # Hessian diagonals of block i that can vanish without making K_i singular (their columns carry constraints)
        done = 0
        for j in range(n_x):
            trial = h1[i].copy()
            trial[j] = 0.0
            Kd = sp.bmat([[sp.diags(trial), Js[i].T], [Js[i], None]]).toarray()
            sv = np.linalg.svd(Kd, compute_uv=False)
            if Js[i].tocsc()[:, j].nnz > 0 and sv.min() > 1e-3 * sv.max():
                h1[i] = trial
                done += 1
                if done == count:
                    return
        raise AssertionError('test setup: no suitable Hessian diagonal in block %d' % i)

    vanish(2, 2)
    vanish(4, 1)
    solver = new_solver(make_engine, nb)
    A0, A1 = kkt(h0), kkt(h1)
    assert solver.do_symbolic_factorization(A0).status == LinearSolverStatus.successful
    for A, refreshes in ((A0, 0), (A1, 1), (A1, 1), (A0, 1)):
        full = dense(A)
        ev = np.linalg.eigvalsh(full)
        assert np.abs(ev).min() > 1e-8 * np.abs(ev).max()            # the matrix itself is regular
        res = solver.do_numeric_factorization(A, raise_on_error=False)
        assert res.status == LinearSolverStatus.successful
        assert solver.pivot_order_refreshes == refreshes
        assert solver.get_inertia() == (int((ev > 0).sum()), int((ev < 0).sum()), 0)
        x = solver.do_back_solve(rhs)
        x_ref = np.linalg.solve(full, rhs.flatten())
        assert np.abs(x.flatten() - x_ref).max() <= 1e-8 * np.abs(x_ref).max()
    # a singular block is still reported as singular (after one attempt with a fresh order)
    h2 = [h.copy() for h in h0]
    A2 = kkt(h2)
    K = A2.get_block(1, 1).tocoo()
    A2.set_block(1, 1, coo_matrix((np.zeros(K.nnz), (K.row, K.col)), shape=K.shape))
    assert solver.do_numeric_factorization(A2, raise_on_error=False).status == LinearSolverStatus.singular


def case_conflicting_pivots(make_engine, nb=6):
    """Two instances of ONE pattern group that need incompatible static pivot sequences (MA27 pivots every block on its
    own values, ma27_interface.py:110-140).  K = [[h1, 0, j1], [0, h2, j2], [j1, j2, 0]] (+ a well-conditioned tail so that
    the blocks are not tiny): type A has h1 = 0, j1 = 1, j2 = 0, h2 = 1 -- the constraint row must pair with x1 --, type
    B has h2 = 0, j2 = 1, j1 = 0, h1 = 1 -- it must pair with x2, and x1 is a plain pivot.  The sequence planned from an A
    block meets an exactly singular 2 x 2 pivot in a B block and vice versa.  Expected: status successful (the group is
    split into two variants), x against a dense solve, exact inertia."""
    rng = np.random.default_rng(3)
    n_t = 5                                     # tail variables (diagonal + a coupling to x1, x2)
    n = 3 + n_t
    nc = 2
    rows = [0, 1, 2, 2, 2] + list(range(3, n)) + [3, 4]
    cols = [0, 1, 0, 1, 2] + list(range(3, n)) + [0, 1]

    def block(kind, i):
        h1, h2, j1, j2 = (0.0, 1.0, 1.0, 0.0) if kind == 'A' else (1.0, 0.0, 0.0, 1.0)
        tail = 2.0 + rng.random(n_t) + 0.1 * i
        vals = [h1, h2, j1, j2, 0.0] + list(tail) + [0.0, 0.0]      # (x1, x2 and the constraint row get nothing from the tail)
        r, c, v = np.array(rows), np.array(cols), np.array(vals)
        off = r != c                              # (both triangles, explicit zeros kept: one pattern for both types)
        return coo_matrix((np.concatenate([v, v[off]]), (np.concatenate([r, c[off]]), np.concatenate([c, r[off]]))), shape=(n, n))

    kinds = ['A', 'B', 'A', 'B', 'B', 'A'][:nb]
    A = BlockMatrix(nb + 1, nb + 1)
    Bs = coo_matrix((np.array([1.0, -1.0]), (np.array([0, 1]), np.array([3, 5]))), shape=(nc, n))
    for i, kd in enumerate(kinds):
        A.set_block(i, i, block(kd, i))
        A.set_block(nb, i, Bs)
        A.set_block(i, nb, Bs.T.tocoo())
    A.set_block(nb, nb, coo_matrix((np.array([4.0, 5.0]), (np.arange(nc), np.arange(nc))), shape=(nc, nc)))
    rhs = BlockVector(nb + 1)
    for i in range(nb):
        rhs.set_block(i, rng.normal(size=n))
    rhs.set_block(nb, rng.normal(size=nc))
    full = A.toarray()
    ev = np.linalg.eigvalsh(full)
    assert np.abs(ev).min() > 1e-8 * np.abs(ev).max()
    solver = new_solver(make_engine, nb)
    assert solver.do_symbolic_factorization(A).status == LinearSolverStatus.successful
    assert len(solver.plan_stats) == 1                                # one pattern group
    for it in range(2):
        res = solver.do_numeric_factorization(A, raise_on_error=False)
        assert res.status == LinearSolverStatus.successful
        assert solver.get_inertia() == (int((ev > 0).sum()), int((ev < 0).sum()), 0)
        x = solver.do_back_solve(rhs)
        x_ref = np.linalg.solve(full, rhs.flatten())
        assert np.abs(x.flatten() - x_ref).max() <= 1e-8 * np.abs(x_ref).max()
        assert scaled_residual(A.tocoo(), x.flatten(), rhs.flatten()) <= RESID_TOL
    assert solver.group_splits >= 1 and len(solver.plan_stats) == 2   # two variants of the pattern, nothing reported singular
    splits = solver.group_splits
    # the split persists: the next factorisation needs no repair
    assert solver.do_numeric_factorization(A, raise_on_error=False).status == LinearSolverStatus.successful
    assert solver.group_splits == splits
    # with the fallback off the caller is told `singular` (what round 4 did)
    solver2 = new_solver(make_engine, nb)
    solver2.split_conflicting_groups = False
    solver2.do_symbolic_factorization(A)
    assert solver2.do_numeric_factorization(A, raise_on_error=False).status == LinearSolverStatus.singular
    return solver


class RecordingTimer(object):
    """Stand-in for pyomo.common.timing.HierarchicalTimer: start/stop by label, must nest and balance."""

    def __init__(self):
        self.stack, self.seen = [], []

    def start(self, name):
        self.stack.append(name)
        self.seen.append(name)

    def stop(self, name):
        assert self.stack and self.stack[-1] == name, (name, self.stack)
        self.stack.pop()


def case_ip_solve_call_pattern(make_engine):
    """The call sites of the reference (interior_point.py:634-652 try_factorization_and_reallocation, :364-400 the
    inertia-correction loop, :566 the back-solve) drive the solver through keywords -- ``matrix=kkt,
    raise_on_error=False, timer=timer`` -- act on ``.status``, ``get_inertia()`` and
    ``increase_memory_allocation()``, and regularise the matrix between retries.  Restated here as the test driver:
    an indefinite Hessian with the wrong inertia must come out factorised with exactly n_con negative eigenvalues
    after a few retries, and the step must solve the regularised system."""
    rng = np.random.default_rng(21)
    n_x, n_c, nc, nb = 9, 3, 2, 4
    Hs, Js, Bs = [], [], []
    for i in range(nb):
        h = rng.uniform(0.5, 2.0, size=n_x)
        h[:2] = -rng.uniform(0.5, 1.0, size=2)                   # negative curvature: wrong inertia without regularisation
        J = (sp.random(n_c, n_x, density=0.4, random_state=50 + i, data_rvs=lambda k: rng.normal(size=k)) +
             2.0 * sp.eye(n_c, n_x)).tocoo()
        B = coo_matrix((rng.normal(size=nc), (np.arange(nc), 2 + rng.choice(n_x - 2, nc, replace=False))),
                       shape=(nc, n_x + n_c))
        Hs.append(h); Js.append(J); Bs.append(B)

    def kkt(dw, dc):
        A = BlockMatrix(nb + 1, nb + 1)
        for i in range(nb):
            H = sp.diags(Hs[i] + dw)
            A.set_block(i, i, sp.bmat([[H, Js[i].T], [Js[i], -dc * sp.eye(n_c)]]).tocoo())
            A.set_block(nb, i, Bs[i])
        A.set_block(nb, nb, (dw * sp.eye(nc)).tocoo())
        return A

    rhs = BlockVector(nb + 1)
    for i in range(nb):
        rhs.set_block(i, rng.normal(size=n_x + n_c))
    rhs.set_block(nb, rng.normal(size=nc))
    solver = new_solver(make_engine, nb)
    timer = RecordingTimer()

    def try_factorization_and_reallocation(matrix, symbolic_or_numeric, max_iter=5):
        method = solver.do_numeric_factorization if symbolic_or_numeric == 'numeric' else \
            solver.do_symbolic_factorization
        for count in range(max_iter):
            res = method(matrix=matrix, raise_on_error=False, timer=timer)
            if res.status == LinearSolverStatus.not_enough_memory:
                solver.increase_memory_allocation(2)
            else:
                break
        return res.status, count

    A = kkt(0.0, 0.0)
    status, _ = try_factorization_and_reallocation(A, 'symbolic')
    assert status == LinearSolverStatus.successful
    status, _ = try_factorization_and_reallocation(A, 'numeric')
    assert status in (LinearSolverStatus.successful, LinearSolverStatus.singular)
    n_con = nb * n_c                                            # n_eq_constraints + n_ineq_constraints
    coef, final, retries = 1e-2, 0.0, 0
    while final <= 1e6:
        pos = neg = zero = None
        if status == LinearSolverStatus.successful:
            pos, neg, zero = solver.get_inertia()
        if neg == n_con and zero == 0 and status == LinearSolverStatus.successful:
            break
        A = kkt(coef, coef)                                     # regularize_equality_gradient(-coef) + regularize_hessian(coef)
        status, _ = try_factorization_and_reallocation(A, 'numeric')
        final, coef, retries = coef, coef * 10.0, retries + 1
    assert retries >= 1 and neg == n_con and zero == 0 and pos == nb * n_x + nc
    x = solver.do_back_solve(rhs, timer=timer)
    m = n_x + n_c
    full = np.zeros((nb * m + nc, nb * m + nc))
    for i in range(nb):
        full[i * m:(i + 1) * m, i * m:(i + 1) * m] = A.get_block(i, i).toarray()
        Bd = A.get_block(nb, i).toarray()
        full[nb * m:, i * m:(i + 1) * m] = Bd
        full[i * m:(i + 1) * m, nb * m:] = Bd.T
    full[nb * m:, nb * m:] = A.get_block(nb, nb).toarray()
    ev = np.linalg.eigvalsh(full)
    assert int((ev < 0).sum()) == n_con
    x_ref = np.linalg.solve(full, rhs.flatten())
    assert np.abs(x.flatten() - x_ref).max() <= 1e-8 * np.abs(x_ref).max()
    assert not timer.stack and {'form SC', 'factorize', 'communicate', 'factor SC', 'back_solve'} <= set(timer.seen)


def case_errors(make_engine):
    import pytest
    # non-square block structure -> ValueError (mpi_...:193-195)
    solver = new_solver(make_engine, 1)
    bad = BlockMatrix(2, 3)
    with pytest.raises(ValueError):
        solver.do_symbolic_factorization(bad)
    # inertia / back-solve before numeric -> RuntimeError (ma27_interface.py:197-200)
    A, rhs = build_8x8(False, 0.0)
    solver = new_solver(make_engine, 3)
    solver.do_symbolic_factorization(A)
    with pytest.raises(RuntimeError):
        solver.get_inertia()
    # singular block: status when raise_on_error=False, RuntimeError otherwise (mpi_...:301-306)
    A.set_block(1, 1, coo_matrix(np.array([[1.0, 1.0], [1.0, 1.0]])))
    solver.do_symbolic_factorization(A)
    res = solver.do_numeric_factorization(A, raise_on_error=False)
    assert res.status == LinearSolverStatus.singular
    assert solver.get_inertia()[2] >= 1
    with pytest.raises(RuntimeError):
        solver.do_numeric_factorization(A, raise_on_error=True)
    # symbolic on a zero-valued copy (quirk Q8), then numeric with the real values
    A, rhs = build_8x8(False, 0.0)
    Z = BlockMatrix(4, 4)
    for i in range(4):
        for j in range(4):
            blk = A.get_block(i, j)
            if blk is not None:
                z = blk.copy()
                z.data[:] = 0.0
                Z.set_block(i, j, z)
    solver = new_solver(make_engine, 3)
    solver.do_symbolic_factorization(Z)
    solver.do_numeric_factorization(A)
    x = solver.do_back_solve(rhs)
    full = np.zeros((8, 8))
    for i in range(3):
        full[2 * i:2 * i + 2, 2 * i:2 * i + 2] = A.get_block(i, i).toarray()
        full[6:, 2 * i:2 * i + 2] = A.get_block(3, i).toarray()
        full[2 * i:2 * i + 2, 6:] = A.get_block(3, i).toarray().T
    assert np.allclose(full @ x.flatten(), rhs.flatten())


def case_singular_schur_complement(make_engine):
    """Quirk Q3: every block regular, S = Q - sum A K^-1 A^T exactly zero.  The reference's MPI class raises here whatever
    the caller asked for (mpi_...:358), its serial class passes the flag (explicit_...:127); this class returns the status
    with raise_on_error=False (what the inertia-correction loop needs) and raises otherwise."""
    import pytest
    A = BlockMatrix(3, 3)
    for i in range(2):
        A.set_block(i, i, coo_matrix(np.array([[2.0]])))
        A.set_block(2, i, coo_matrix(np.array([[1.0]])))
    A.set_block(2, 2, coo_matrix(np.array([[1.0]])))
    solver = new_solver(make_engine, 2)
    assert solver.do_symbolic_factorization(A).status == LinearSolverStatus.successful
    res = solver.do_numeric_factorization(A, raise_on_error=False)
    assert res.status == LinearSolverStatus.singular
    with pytest.raises(RuntimeError):
        solver.do_numeric_factorization(A, raise_on_error=True)
    A.set_block(2, 2, coo_matrix(np.array([[3.0]])))          # S = 2: regular again, same object
    assert solver.do_numeric_factorization(A).status == LinearSolverStatus.successful
    rhs = BlockVector(3)
    for i, v in enumerate((1.0, 2.0, 3.0)):
        rhs.set_block(i, np.array([v]))
    x = solver.do_back_solve(rhs).flatten()
    full = np.array([[2.0, 0, 1], [0, 2.0, 1], [1, 1, 3.0]])
    assert np.allclose(full @ x, rhs.flatten())
    assert solver.get_inertia() == (3, 0, 0)


def case_nested_blocks(make_engine):
    """Quirk Q9: diagonal blocks that are nested BlockMatrix objects ([[H, J^T], [J, -D]], as sc_ip_interface.py hands them
    over) and right-hand-side blocks that are nested BlockVectors give the results of the flat form, in the structure of the
    right-hand side (mpi_...:384, 395; ma27_interface.py:169-182)."""
    N = 3
    model = SyntheticKKT(N, 6, 2, 3)
    comm = SerialComm()
    flat_kkt = model.build_kkt(comm=comm, iteration=1)
    flat_rhs = model.build_rhs(comm=comm)
    rng = np.random.default_rng(5)
    for ndx in range(N + 1):
        flat_rhs.set_block(ndx, rng.standard_normal(flat_rhs.get_block(ndx).size))
    ny = model.n_y
    nested = BlockMatrix(N + 1, N + 1)
    nrhs = BlockVector(N + 1)
    for ndx in range(N):
        K = flat_kkt.get_block(ndx, ndx).tocsr()
        n = K.shape[0]
        kb = BlockMatrix(2, 2)
        kb.set_block(0, 0, K[:ny, :ny].tocoo())
        kb.set_block(0, 1, K[:ny, ny:].tocoo())
        kb.set_block(1, 0, K[ny:, :ny].tocoo())
        kb.set_block(1, 1, K[ny:, ny:].tocoo())
        nested.set_block(ndx, ndx, kb)
        Ab = flat_kkt.get_block(N, ndx).tocsr()
        ab = BlockMatrix(1, 2)
        ab.set_block(0, 0, Ab[:, :ny].tocoo())
        ab.set_block(0, 1, Ab[:, ny:].tocoo())
        nested.set_block(N, ndx, ab)
        v = flat_rhs.get_block(ndx)
        vb = BlockVector(2)
        vb.set_block(0, v[:ny].copy())
        vb.set_block(1, v[ny:].copy())
        nrhs.set_block(ndx, vb)
        assert n == model.block_dim
    Q = flat_kkt.get_block(N, N)
    if Q is not None:
        nested.set_block(N, N, Q)
    cb = BlockVector(1)
    cb.set_block(0, flat_rhs.get_block(N).copy())
    nrhs.set_block(N, cb)
    xs = []
    for kkt, rhs in ((flat_kkt, flat_rhs), (nested, nrhs)):
        solver = new_solver(make_engine, N)
        assert solver.do_symbolic_factorization(kkt).status == LinearSolverStatus.successful
        assert solver.do_numeric_factorization(kkt).status == LinearSolverStatus.successful
        xs.append(solver.do_back_solve(rhs))
    xf, xn = xs
    assert np.array_equal(xf.flatten(), xn.flatten())
    F = flat_kkt.toarray()
    nK = N * model.block_dim
    F[:nK, nK:] = F[nK:, :nK].T
    assert scaled_residual(F, xn.flatten(), flat_rhs.flatten()) <= 1e-10
    for ndx in range(N):
        blk = xn.get_block(ndx)
        assert hasattr(blk, 'get_block') and blk.nblocks == 2 and blk.get_block(0).size == ny
    assert hasattr(xn.get_block(N), 'get_block') and xn.get_block(N).nblocks == 1
    assert not hasattr(xf.get_block(0), 'get_block')
    # the nested right-hand side is left as it was
    assert np.array_equal(nrhs.flatten(), flat_rhs.flatten())


# ---- memory reallocation protocol (linalg/tests/test_realloc.py:10-61, interior_point.py:634-652) ----------
def case_reallocation(make_engine, required_bytes):
    """A solver whose device-storage budget is too small reports not_enough_memory from the numeric phase (status,
    not an exception, when raise_on_error=False); the caller's loop grows the allocation and tries again."""
    N = 6
    model = SyntheticKKT(N, 40, 2, 8)
    kkt = model.build_kkt(comm=SerialComm(), iteration=1)
    rhs = model.build_rhs(comm=SerialComm())
    probe = new_solver(make_engine, N)
    probe.do_symbolic_factorization(kkt)
    need = required_bytes(probe)
    assert need > 0
    solver = new_solver(make_engine, N, memory_budget_bytes=max(1, need // 3))
    assert solver.do_symbolic_factorization(matrix=kkt, raise_on_error=False).status == LinearSolverStatus.successful
    res = solver.do_numeric_factorization(matrix=kkt, raise_on_error=False)
    assert res.status == LinearSolverStatus.not_enough_memory
    import pytest
    with pytest.raises(RuntimeError, match='not_enough_memory'):
        solver.do_numeric_factorization(kkt)
    for count in range(5):                              # try_factorization_and_reallocation, max_iter = 5, factor 2
        res = solver.do_numeric_factorization(matrix=kkt, raise_on_error=False)
        if res.status == LinearSolverStatus.not_enough_memory:
            solver.increase_memory_allocation(2)
        else:
            break
    assert res.status == LinearSolverStatus.successful and count == 2
    x = solver.do_back_solve(rhs)
    assert scaled_residual(kkt.tocoo(), x.flatten(), rhs.flatten()) <= RESID_TOL
    n_y = model.n_y
    assert solver.get_inertia() == (N * (n_y + model.n_q) + 8, N * (n_y + 8), 0)
    return need


def reference_realloc_matrix(n=10000):
    """linalg/tests/test_realloc.py:17-33: tridiagonal, 1e-7 on the diagonal, 1e2 beside it -- a matrix that needs a 2x2
    pivot everywhere."""
    small_val, big_val = 1e-7, 1e2
    irn, jcn, ent = [], [], []
    for i in range(n - 1):
        irn.extend([i + 1, i, i])
        jcn.extend([i, i, i + 1])
        ent.extend([big_val, small_val, big_val])
    irn.append(n - 1)
    jcn.append(n - 1)
    ent.append(small_val)
    from scipy.sparse import coo_matrix
    return coo_matrix((np.array(ent), (np.array(irn), np.array(jcn))), shape=(n, n))


def case_reference_realloc_matrix(make_engine, n=10000):
    """The reference's reallocation fixture (linalg/tests/test_realloc.py:10-61) through the sub-solver adapters: a
    budget below what the plan needs -> not_enough_memory (status, then exception) -> increase_memory_allocation(2) ->
    success; solution against SuperLU, inertia (n/2, n/2, 0) -- every pivot is a 2x2 one."""
    import pytest
    from scipy.sparse.linalg import splu
    from parapint_amd.linalg import HipLDLInterface, MumpsInterface, ScipyInterface, InteriorPointMA27Interface
    assert InteriorPointMA27Interface is HipLDLInterface
    matrix = reference_realloc_matrix(n)
    rng = np.random.default_rng(0)
    b = rng.normal(size=n)
    x_ref = splu(matrix.tocsc()).solve(b)
    probe = MumpsInterface(engine=make_engine())
    probe.do_symbolic_factorization(matrix)
    need = probe._sc._eng.memory_info()[0]
    assert need > 0
    for cls, kw in ((MumpsInterface, dict(par=1, comm=None, cntl_options={1: 0.01}, icntl_options={14: 20})),
                    (HipLDLInterface, dict(cntl_options={1: 1e-6}))):
        solver = cls(engine=make_engine(), **kw)
        solver._sc._eng.set_memory_budget(max(1, int(0.6 * need)))
        solver.do_symbolic_factorization(matrix)
        res = solver.do_numeric_factorization(matrix, raise_on_error=False)
        assert res.status == LinearSolverStatus.not_enough_memory
        with pytest.raises(RuntimeError, match='not_enough_memory'):
            solver.do_numeric_factorization(matrix)
        solver.do_symbolic_factorization(matrix)
        solver.increase_memory_allocation(2)
        res = solver.do_numeric_factorization(matrix)
        assert res.status == LinearSolverStatus.successful
        x = solver.do_back_solve(b)
        assert np.abs(x - x_ref).max() <= 1e-8 * np.abs(x_ref).max()
        assert solver.get_inertia() == (n // 2, n // 2, 0)
        if cls is MumpsInterface:
            assert solver.get_infog(12) == n // 2 and solver.get_infog(28) == 0
            assert solver.get_icntl(13) == 1 and solver.get_icntl(24) == 0 and solver.get_icntl(14) == 20
            with pytest.raises(ValueError):
                solver.set_icntl(24, 1)
    sp = ScipyInterface(engine=make_engine())
    sp.do_symbolic_factorization(matrix)
    sp.do_numeric_factorization(matrix)
    assert np.abs(sp.do_back_solve(b) - x_ref).max() <= 1e-8 * np.abs(x_ref).max()
    with pytest.raises(RuntimeError, match='compute_inertia'):
        sp.get_inertia()
    sp2 = ScipyInterface(compute_inertia=True, engine=make_engine())
    sp2.do_symbolic_factorization(matrix)
    sp2.do_numeric_factorization(matrix)
    assert sp2.get_inertia() == (n // 2, n // 2, 0)


def case_status_severity():
    """One reduction must let the most severe status win: `warning` (enum value 4) may not mask `singular` (2)."""
    from parapint_amd.linalg import hip_schur_complement as mod
    order = [LinearSolverStatus.successful, LinearSolverStatus.warning, LinearSolverStatus.not_enough_memory,
             LinearSolverStatus.singular, LinearSolverStatus.error]
    ranks = [mod._SEVERITY[st] for st in order]
    assert ranks == sorted(ranks) and len(set(ranks)) == 5
    for st in order:
        assert mod._BY_SEVERITY[mod._SEVERITY[st]] == st


# ---- pivot-growth guard (MA27 cntl(1): ma27_interface.py:36-47, examples/stochastic.py:120-124) ----------------------
def case_growth_guard(make_engine):
    """The pivot sequence is static per pattern group.  A later matrix in which ONE instance has a Hessian diagonal that is
    1e-9 of its column (not zero: the zero-pivot rule does not fire) is factorised with an element growth of 1e9 in that
    instance.  Without a tolerance the factorisation succeeds and only counts the instance; with ``pivot_tolerance`` u the
    growth |l_ij| > 1/u is treated like a breakdown: the order is refreshed from that instance (a 2x2 pivot takes the weak
    diagonal) and the solve is accurate to 1e-8 again."""
    rng = np.random.default_rng(17)
    n_x, n_c, nc, nb = 8, 3, 2, 5
    Js, Bs = [], []
    for i in range(nb):
        J = (sp.random(n_c, n_x, density=0.4, random_state=70 + i, data_rvs=lambda k: rng.normal(size=k)) +
             2.0 * sp.eye(n_c, n_x)).tocoo()
        B = coo_matrix((rng.normal(size=nc), (np.arange(nc), rng.choice(n_x, nc, replace=False))), shape=(nc, n_x + n_c))
        Js.append(J); Bs.append(B)
    h0 = [rng.uniform(1.0, 3.0, size=n_x) for _ in range(nb)]
    h1 = [h.copy() for h in h0]
    weak = None
    for j in range(n_x):                               # a Hessian diagonal of block 3 that can nearly vanish safely
        trial = h1[3].copy()
        trial[j] = 1e-9
        Kd = sp.bmat([[sp.diags(trial), Js[3].T], [Js[3], None]]).toarray()
        sv = np.linalg.svd(Kd, compute_uv=False)
        if Js[3].tocsc()[:, j].nnz > 0 and sv.min() > 1e-3 * sv.max():
            h1[3], weak = trial, j
            break
    assert weak is not None

    def kkt(hs):
        A = BlockMatrix(nb + 1, nb + 1)
        for i in range(nb):
            H = coo_matrix((hs[i], (np.arange(n_x), np.arange(n_x))), shape=(n_x, n_x))
            A.set_block(i, i, sp.bmat([[H, Js[i].T], [Js[i], None]]).tocoo())
            A.set_block(nb, i, Bs[i])
        A.set_block(nb, nb, coo_matrix((nc, nc)))
        return A

    m = n_x + n_c
    A0, A1 = kkt(h0), kkt(h1)
    full = np.zeros((nb * m + nc, nb * m + nc))
    for i in range(nb):
        Kd = A1.get_block(i, i).toarray()
        full[i * m:(i + 1) * m, i * m:(i + 1) * m] = np.tril(Kd) + np.tril(Kd, -1).T
        Bd = A1.get_block(nb, i).toarray()
        full[nb * m:, i * m:(i + 1) * m] = Bd
        full[i * m:(i + 1) * m, nb * m:] = Bd.T
    rhs = BlockVector(nb + 1)
    for i in range(nb):
        rhs.set_block(i, rng.normal(size=m))
    rhs.set_block(nb, rng.normal(size=nc))
    x_ref = np.linalg.solve(full, rhs.flatten())
    ev = np.linalg.eigvalsh(full)
    # (1) no tolerance: accepted silently, the instance is counted
    plain = new_solver(make_engine, nb)
    plain.do_symbolic_factorization(A0)
    assert plain.do_numeric_factorization(A1, raise_on_error=False).status == LinearSolverStatus.successful
    assert plain.pivot_order_refreshes == 0 and plain.growth_instances == 1
    # (2) MA27's pivot tolerance given: growth beyond 1/u refreshes the order, then the factorisation is clean
    guarded = new_solver(make_engine, nb, pivot_tolerance=1e-6)
    try:
        guarded.do_symbolic_factorization(A0)
        assert guarded.do_numeric_factorization(A0, raise_on_error=False).status == LinearSolverStatus.successful
        assert guarded.pivot_order_refreshes == 0
        res = guarded.do_numeric_factorization(A1, raise_on_error=False)
        assert res.status == LinearSolverStatus.successful
        assert guarded.pivot_order_refreshes == 1 and guarded.growth_instances == 0
        assert guarded.get_inertia() == (int((ev > 0).sum()), int((ev < 0).sum()), 0)
        x = guarded.do_back_solve(rhs)
        assert np.abs(x.flatten() - x_ref).max() <= 1e-8 * np.abs(x_ref).max()
    finally:
        if hasattr(guarded._eng, 'set_pivot_tolerance'):
            guarded._eng.set_pivot_tolerance(0.0, 0.0)       # (the host interpreter keeps the tolerance process-wide)


# ---- dynamic (time-staged) problems: block-banded S (sc_ip_interface.py:274-357, mpi_...:88-125, 228-255) ------------
def case_dynamic(make_engine, T, n_s, n_u=2, nfe=3, iteration=1, expect_block_tridiagonal=None, oracle=True, comm=None,
                 dense_limit=None):
    """The KKT layout of the reference's dynamic interface on a synthetic linear-quadratic problem: every time block
    touches only the coupling rows of its own two links, so the blocks of one pattern share a plan through LOCAL
    coupling rows + per-block maps and S is assembled by scattering their cliques -- dense for small systems,
    block tridiagonal (after a bandwidth-reducing ordering) for large ones.  Against the oracle's restatement of the
    reference (sparse S pattern from the union of the cliques), dense algebra where affordable, residual and inertia."""
    from parapint_amd.examples.performance.schur_complement.dynamic_kkt import SyntheticDynamicKKT
    model = SyntheticDynamicKKT(T, n_s, n_u, nfe)
    kkt = model.build_kkt(comm=SerialComm() if comm is None else comm, iteration=iteration)
    rhs = model.build_rhs(comm=SerialComm() if comm is None else comm)
    solver = new_solver(make_engine, T, comm=comm)
    if dense_limit is not None:
        solver._dense_coupling_limit = dense_limit
    assert solver.do_symbolic_factorization(kkt).status == LinearSolverStatus.successful
    assert len(solver.plan_stats) == min(T, 3) or comm is not None                 # first, interior and last time blocks: three plans
    if expect_block_tridiagonal is not None:
        assert (solver._btd is not None) == expect_block_tridiagonal
    assert solver.do_numeric_factorization(kkt).status == LinearSolverStatus.successful
    x = solver.do_back_solve(rhs)
    K = kkt.tocoo()
    assert scaled_residual(K, x.flatten(), rhs.flatten()) <= RESID_TOL
    n_vars = T * model.n_x + n_s * (T - 1)
    n_cons = T * model.n_eq + 2 * n_s * (T - 1)
    assert solver.get_inertia() == (n_vars, n_cons, 0)
    if oracle:
        oc = OracleMPISC({i: OracleScipy() for i in range(T)}, OracleScipy())
        oc.do_symbolic_factorization(kkt)
        oc.do_numeric_factorization(kkt)
        xo = oc.do_back_solve(rhs).flatten()
        assert np.abs(x.flatten() - xo).max() <= 1e-8 * np.abs(xo).max()
        So = sp.coo_matrix(oc.schur_complement).toarray()
        S = solver.get_schur_complement()
        S = S.toarray() if sp.issparse(S) else S
        assert np.abs(S - So).max() <= 1e-9 * np.abs(So).max()
        assert oc.schur_complement.nnz < So.size or T <= 2       # the reference's S pattern is sparse, too
    if K.shape[0] <= 3000:
        xd = np.linalg.solve(K.toarray(), rhs.flatten())
        assert np.abs(x.flatten() - xd).max() <= 1e-8 * np.abs(xd).max()
    # a second factorisation with other values on the same plan
    kkt2 = model.build_kkt(comm=SerialComm() if comm is None else comm, iteration=iteration + 1)
    assert solver.do_numeric_factorization(kkt2).status == LinearSolverStatus.successful
    x2 = solver.do_back_solve(rhs)
    assert scaled_residual(kkt2.tocoo(), x2.flatten(), rhs.flatten()) <= RESID_TOL
    return solver, model


def case_dynamic_regularised(make_engine, dense_limit=None):
    """The inertia-correction loop on a dynamic problem (interior_point.py:364-392 with sc_ip_interface.py:903-933): the
    regularised KKT has delta on the link-dual diagonals of every time block and on both halves of the coupling block --
    entries outside the planned pattern -> one re-plan on the union pattern (the structure of S is found again, with the
    new diagonal of Q), then the retries and the next unregularised matrix factorise on that plan."""
    import scipy.sparse as sp2
    from parapint_amd.examples.performance.schur_complement.dynamic_kkt import SyntheticDynamicKKT
    T, n_s = 10, 3
    model = SyntheticDynamicKKT(T, n_s, 2, 2)
    rhs = model.build_rhs(comm=SerialComm())
    solver = new_solver(make_engine, T)
    if dense_limit is not None:
        solver._dense_coupling_limit = dense_limit
    kkt = model.build_kkt(comm=SerialComm(), iteration=1)
    solver.do_symbolic_factorization(kkt)
    assert solver.do_numeric_factorization(kkt).status == LinearSolverStatus.successful

    def regularised(coef):
        R = model.build_kkt(comm=SerialComm(), iteration=1)
        for t in range(T):
            blk = R.get_block(t, t)                                   # [[kkt_t, L^T], [L, 0 * I]]
            inner = blk.get_block(0, 0)
            n_x, n_eq = model.n_x, model.n_eq
            inner.set_block(0, 0, (inner.get_block(0, 0) + coef * sp2.identity(n_x, format='coo')).tocoo())
            inner.set_block(1, 1, (-coef * sp2.identity(n_eq, format='coo')).tocoo())
            nb = blk.get_block(1, 1).shape[0]
            blk.set_block(1, 1, (-coef * sp2.identity(nb, format='coo')).tocoo())
        R.set_block(T, T, (model.corner_matrix() + coef * sp2.identity(model.n_coupling, format='coo')).tocoo())
        return R
    for coef in (1e-6, 1e-3):
        R = regularised(coef)
        res = solver.do_numeric_factorization(R, raise_on_error=False)
        assert res.status == LinearSolverStatus.successful
        K = R.tocoo().toarray()
        ev = np.linalg.eigvalsh(K)
        assert solver.get_inertia() == (int((ev > 0).sum()), int((ev < 0).sum()), 0)
        x = solver.do_back_solve(rhs)
        xd = np.linalg.solve(K, rhs.flatten())
        assert np.abs(x.flatten() - xd).max() <= 1e-8 * np.abs(xd).max()
    groups_before = [id(g) for g in solver._groups]
    kkt2 = model.build_kkt(comm=SerialComm(), iteration=2)
    assert solver.do_numeric_factorization(kkt2).status == LinearSolverStatus.successful
    assert [id(g) for g in solver._groups] == groups_before          # a subset of the union pattern: no new plan
    x = solver.do_back_solve(rhs)
    assert scaled_residual(kkt2.tocoo(), x.flatten(), rhs.flatten()) <= RESID_TOL
    return solver


# ---- host boundary: which blocks take which way to the device --------------------------------------
def case_constant_entries(make_engine, calls=None):
    """declare_constant_entries: the producer names the entries that do not change between numeric factorisations (all but
    the Hessian diagonal of the synthetic KKT system); later factorisations look at the others only and give the results
    of an undeclared solver; a declaration that does not hold is reported with check=True (the values handed over are used
    all the same) and heals at every `pattern_check_interval`-th call otherwise; re-plans and a new symbolic phase keep / end
    the declaration."""
    N = 6
    model = SyntheticKKT(N, 3, 8, 2)
    comm = SerialComm()
    rhs = model.build_rhs(comm=comm)
    rng = np.random.default_rng(11)
    for ndx in range(N):
        rhs.set_block(ndx, rng.standard_normal(model.block_dim))
    plain = new_solver(make_engine, N)
    decl = new_solver(make_engine, N)
    eng_calls = getattr(decl._eng, 'calls', None) if calls is None else calls
    try:
        decl.declare_constant_entries(model.constant_entries())
        raise AssertionError('a declaration before the symbolic phase must be refused')
    except RuntimeError:
        pass
    kkt = model.build_kkt(comm=comm, iteration=0)
    for s in (plain, decl):
        s.do_symbolic_factorization(kkt)
    decl.declare_constant_entries(model.constant_entries())
    assert all(g.var_runs is not None for g in decl._groups)
    nvar = sum(int(g.var_runs[0][:, 1].sum() + g.var_runs[1][:, 1].sum()) for g in decl._groups)
    assert model.n_y <= nvar < model.nnz_per_block // 2                # (the diagonal plus the gaps closed between its runs)
    for it in range(0, 5):
        k = model.build_kkt(comm=comm, iteration=it)
        before = dict(eng_calls) if eng_calls is not None else None
        for s in (plain, decl):
            assert s.do_numeric_factorization(k).status == LinearSolverStatus.successful
        xp, xd = plain.do_back_solve(rhs), decl.do_back_solve(rhs)
        assert np.array_equal(xp.flatten(), xd.flatten())
        assert scaled_residual(k.toarray(), xd.flatten(), rhs.flatten()) <= 1e-10
        if eng_calls is not None and it >= 2 and 'variable_entries' in eng_calls:
            # the declared solver touched the variable entries only (rows staged whole at the first verified pass)
            assert eng_calls['variable_entries'] - before.get('variable_entries', 0) == N * nvar
    # a declaration that does not hold: a Jacobian entry of block 2 changes
    k = model.build_kkt(comm=comm, iteration=7)
    K2 = k.get_block(2, 2)
    K2.data = K2.data.copy()
    def pair(q):
        # an entry below the diagonal (read, declared constant) and its mirror image above (the check is of the full matrix)
        e = int(np.flatnonzero(model._row > model._col)[q])
        m = int(np.flatnonzero((model._row == model._col[e]) & (model._col == model._row[e]))[0])
        return [e, m]

    e_const = pair(3)
    K2.data[e_const] *= 1.25
    decl.declare_constant_entries(model.constant_entries(), check=True)
    k_ok = model.build_kkt(comm=comm, iteration=6)
    assert decl.do_numeric_factorization(k_ok).status == LinearSolverStatus.successful
    res = decl.do_numeric_factorization(k, raise_on_error=False)
    assert res.status == LinearSolverStatus.error and 'declared constant' in decl._last_error
    # ... the values were sent all the same: the next call (same matrix) is clean and solves the changed system
    assert decl.do_numeric_factorization(k).status == LinearSolverStatus.successful
    x = decl.do_back_solve(rhs)
    assert scaled_residual(k.toarray(), x.flatten(), rhs.flatten()) <= 1e-10
    # default (check=None): a rotating sample of blocks is compared at every call -- with a sample of six all of them, so the
    # violation is reported at once, and the values handed over are used all the same
    decl.constant_sample_blocks = 6
    decl.declare_constant_entries(model.constant_entries())
    assert decl.do_numeric_factorization(k).status == LinearSolverStatus.successful           # (same values as staged: clean)
    K2.data[e_const] *= 1.1
    res = decl.do_numeric_factorization(k, raise_on_error=False)
    assert res.status == LinearSolverStatus.error and 'declared constant' in decl._last_error
    assert decl.do_numeric_factorization(k).status == LinearSolverStatus.successful
    assert scaled_residual(k.toarray(), decl.do_back_solve(rhs).flatten(), rhs.flatten()) <= 1e-10
    # ... with many blocks per group only `constant_sample_blocks` of them per call (a rotating residue class)
    decl.constant_sample_blocks = 2             # stride 3: blocks {0, 3}, {1, 4}, {2, 5} in turn
    hits = 0
    for rep in range(3):
        K2.data[e_const] *= 1.1
        hits += decl.do_numeric_factorization(k, raise_on_error=False).status == LinearSolverStatus.error
        decl.do_numeric_factorization(k, raise_on_error=False)      # (the call after a report stages block 2 in full: clean again)
    assert 1 <= hits <= 3
    decl.constant_sample_blocks = 2
    # check=False: no comparison -- every `pattern_check_interval`-th call stages every entry again, a declaration that does
    # not hold heals there
    decl.declare_constant_entries(model.constant_entries(), check=False)
    assert decl.do_numeric_factorization(k).status == LinearSolverStatus.successful
    decl.pattern_check_interval = 3
    K2.data[e_const] *= 0.8
    decl._stage_calls = 0                       # (the third call from here is the one that stages every entry)
    worst = []
    for rep in range(3):
        assert decl.do_numeric_factorization(k, raise_on_error=False).status == LinearSolverStatus.successful
        worst.append(scaled_residual(k.toarray(), decl.do_back_solve(rhs).flatten(), rhs.flatten()))
    assert max(worst) > 1e-6 and worst[-1] <= 1e-10, worst        # (stale until the full pass, right from there on)
    decl.pattern_check_interval = 8
    # withdrawn: every entry is looked at again
    decl.declare_constant_entries(None)
    assert all(g.var_runs is None for g in decl._groups)
    K2.data[pair(5)] *= 1.5
    assert decl.do_numeric_factorization(k).status == LinearSolverStatus.successful
    x = decl.do_back_solve(rhs)
    assert scaled_residual(k.toarray(), x.flatten(), rhs.flatten()) <= 1e-10
    # a new symbolic phase ends a declaration
    decl.declare_constant_entries(model.constant_entries())
    decl.do_symbolic_factorization(kkt)
    assert all(g.var_runs is None for g in decl._groups)
    # masks of the wrong length (another problem's) are ignored
    bad = {ndx: (np.ones(5, dtype=bool), None) for ndx in range(N)}
    decl.declare_constant_entries(bad)
    assert all(g.var_runs is None for g in decl._groups)
    assert decl.do_numeric_factorization(kkt).status == LinearSolverStatus.successful
    # a declaration outlives a re-plan only where it still describes the entries: the regularised matrix of an
    # inertia-correction step has entries outside the planned pattern -> one re-plan on the union pattern, whose raw layout
    # is not the declared one any more -> the declaration is dropped for those groups, every entry is read
    decl.declare_constant_entries(model.constant_entries())
    assert all(g.var_runs is not None for g in decl._groups)
    assert decl.do_numeric_factorization(model.build_kkt(comm=comm, iteration=1)).status == LinearSolverStatus.successful
    reg = model.build_kkt(comm=comm, iteration=2)
    n = model.block_dim
    for ndx in range(N):
        K = reg.get_block(ndx, ndx)
        d = np.concatenate([np.zeros(model.n_y + model.n_q), -1e-3 * np.ones(n - model.n_y - model.n_q)])
        reg.set_block(ndx, ndx, (K + sp.diags(d)).tocoo())
    assert decl.do_numeric_factorization(reg).status == LinearSolverStatus.successful
    assert all(g.var_runs is None for g in decl._groups)
    x = decl.do_back_solve(rhs)
    assert scaled_residual(reg.toarray(), x.flatten(), rhs.flatten()) <= 1e-10
    back = model.build_kkt(comm=comm, iteration=3)               # a subset of the union pattern again
    assert decl.do_numeric_factorization(back).status == LinearSolverStatus.successful
    x = decl.do_back_solve(rhs)
    assert scaled_residual(back.toarray(), x.flatten(), rhs.flatten()) <= 1e-10


def case_flat_values(make_engine, calls=None):
    """HostValueMatrix: the pattern object of the symbolic phase + one flat value vector per block (the rows of one 2-D
    array, or a dictionary of vectors).  Same results as the COO blocks with the same values; one staging call per pattern
    group for the 2-D form; works with a declaration of constant entries, after a re-plan on a union pattern (blocks are then
    canonicalised on the host), with a coupling block of its own; a matrix over another pattern object is refused."""
    from parapint_amd.sparse.host_value_matrix import HostValueMatrix
    N = 7
    model = SyntheticKKT(N, 3, 8, 2)
    comm = SerialComm()
    rhs = model.build_rhs(comm=comm)
    rng = np.random.default_rng(3)
    for ndx in range(N):
        rhs.set_block(ndx, rng.standard_normal(model.block_dim))
    ref = new_solver(make_engine, N)
    flat = new_solver(make_engine, N)
    eng_calls = getattr(flat._eng, 'calls', None) if calls is None else calls
    pattern = model.build_kkt(comm=comm, iteration=0)
    ref.do_symbolic_factorization(pattern)
    flat.do_symbolic_factorization(HostValueMatrix(pattern))            # (the wrapped pattern ...)
    assert flat.do_numeric_factorization(HostValueMatrix(pattern)).status == LinearSolverStatus.successful   # ... and its own values
    x = flat.do_back_solve(rhs)
    assert scaled_residual(pattern.toarray(), x.flatten(), rhs.flatten()) <= 1e-10
    for it, form in ((1, '2d'), (2, 'dict'), (3, '2d'), (4, 'strided')):
        kkt = model.build_kkt(comm=comm, iteration=it)
        vals = model.flat_values(iteration=it)
        if form == 'dict':
            vals = {ndx: vals[i].copy() for i, ndx in enumerate(range(N))}
        elif form == 'strided':
            wide = np.zeros((N, vals.shape[1] + 5))
            wide[:, :vals.shape[1]] = vals
            vals = wide[:, :vals.shape[1]]                     # rows contiguous, row stride larger than a row
        before = dict(eng_calls) if eng_calls is not None else None
        assert ref.do_numeric_factorization(kkt).status == LinearSolverStatus.successful
        hv = HostValueMatrix(pattern, vals)
        assert flat.do_numeric_factorization(hv).status == LinearSolverStatus.successful
        xr, xf = ref.do_back_solve(rhs), flat.do_back_solve(rhs)
        assert np.array_equal(xr.flatten(), xf.flatten())
        assert flat.get_inertia() == ref.get_inertia()
        assert np.array_equal(hv.to_block_matrix().toarray(), kkt.toarray())
        if eng_calls is not None and form != 'dict':
            # one staging call for the one pattern group (the reference solver's blocks went one by one or in batches)
            # (the reference solver shares the counters only when `calls` is given: its blocks go in batches of 64)
            assert 1 <= eng_calls['stage_upload_verified'] - before['stage_upload_verified'] <= 2
            assert eng_calls['verified_blocks'] - before['verified_blocks'] in (N, 2 * N)
    # with a declaration of constant entries
    flat.declare_constant_entries(model.constant_entries())
    for it in (5, 6, 7):
        kkt = model.build_kkt(comm=comm, iteration=it)
        ref.do_numeric_factorization(kkt)
        assert flat.do_numeric_factorization(HostValueMatrix(pattern, model.flat_values(iteration=it))).status == \
            LinearSolverStatus.successful
        assert np.array_equal(ref.do_back_solve(rhs).flatten(), flat.do_back_solve(rhs).flatten())
    # a coupling block of its own
    Q = coo_matrix(np.diag([0.5, 0.25]))
    kkt = model.build_kkt(comm=comm, iteration=8)
    kkt.set_block(N, N, Q)
    ref.do_numeric_factorization(kkt)
    flat.do_numeric_factorization(HostValueMatrix(pattern, model.flat_values(iteration=8), Q=Q))
    xf = flat.do_back_solve(rhs)
    assert np.array_equal(ref.do_back_solve(rhs).flatten(), xf.flatten())
    assert scaled_residual(kkt.toarray(), xf.flatten(), rhs.flatten()) <= 1e-10
    # another pattern object, a wrong shape
    other = model.build_kkt(comm=comm, iteration=0)
    for bad in (HostValueMatrix(other, model.flat_values(iteration=1)), HostValueMatrix(pattern, np.zeros((N, 3)))):
        try:
            flat.do_numeric_factorization(bad)
            raise AssertionError('refused matrices must raise')
        except (RuntimeError, ValueError) as err:
            assert 'HostValueMatrix' in str(err) or 'flat_values' in str(err), str(err)
    # after a re-plan on a union pattern (a regularised COO matrix in between) the flat form still works: its blocks are
    # then in another entry order than the new plan's and are canonicalised on the host
    reg = model.build_kkt(comm=comm, iteration=9)
    n = model.block_dim
    for ndx in range(N):
        K = reg.get_block(ndx, ndx)
        d = np.concatenate([np.zeros(model.n_y + model.n_q), -1e-3 * np.ones(n - model.n_y - model.n_q)])
        reg.set_block(ndx, ndx, (K + sp.diags(d)).tocoo())
    assert flat.do_numeric_factorization(reg).status == LinearSolverStatus.successful
    kkt = model.build_kkt(comm=comm, iteration=10)
    assert flat.do_numeric_factorization(HostValueMatrix(pattern, model.flat_values(iteration=10))).status == \
        LinearSolverStatus.successful
    x = flat.do_back_solve(rhs)
    assert scaled_residual(kkt.toarray(), x.flatten(), rhs.flatten()) <= 1e-10


def case_boundary_fast_paths(make_engine, calls=None):
    """do_numeric_factorization / do_back_solve with host blocks, over the ways an interface hands its matrix over:
    new COO blocks over the same index arrays, the same blocks with .data rewritten in place or replaced, index arrays
    rewritten in place (a different entry order for the same matrix), both result-buffer modes.  Every result is
    checked against a dense solve; `calls` (the CPU engine's counters) tells which path the blocks took."""
    N = 5
    model = SyntheticKKT(N, 3, 8, 2)
    comm = SerialComm()
    rng = np.random.default_rng(5)

    def dense_check(solver, kkt, rhs, x):
        K = kkt.toarray()
        b = rhs.flatten()
        assert scaled_residual(K, x.flatten(), b) <= 1e-10

    for buffers in (0, 2):
        solver = new_solver(make_engine, N, result_buffers=buffers)
        eng_calls = getattr(solver._eng, 'calls', None) if calls is None else calls
        kkt = model.build_kkt(comm=comm, iteration=0)
        rhs = model.build_rhs(comm=comm)
        for ndx in range(N):
            rhs.set_block(ndx, rng.standard_normal(model.block_dim))
        solver.do_symbolic_factorization(kkt)
        solver.do_numeric_factorization(kkt)
        x0 = solver.do_back_solve(rhs)
        dense_check(solver, kkt, rhs, x0)
        x0_copy = x0.flatten().copy()
        # new blocks over the same index arrays
        kkt1 = model.build_kkt(comm=comm, iteration=1)
        before = dict(eng_calls) if eng_calls is not None else None
        solver.do_numeric_factorization(kkt1)
        x1 = solver.do_back_solve(rhs)
        dense_check(solver, kkt1, rhs, x1)
        if eng_calls is not None:
            assert eng_calls['verified_blocks'] - before['verified_blocks'] == N          # nothing compared again
            assert eng_calls['compared_blocks'] == before['compared_blocks']
            assert eng_calls['upload_rhs_rows'] > before['upload_rhs_rows']
        if buffers == 0:
            assert np.array_equal(x0.flatten(), x0_copy)       # fresh arrays per call: the earlier result is untouched
        # the same blocks, .data rewritten in place / replaced by a new array
        for ndx in range(N):
            K = kkt1.get_block(ndx, ndx)
            K.data[:] = model.block_values(ndx, 2)
        solver.do_numeric_factorization(kkt1)
        dense_check(solver, kkt1, rhs, solver.do_back_solve(rhs))
        for ndx in range(N):
            kkt1.get_block(ndx, ndx).data = np.array(model.block_values(ndx, 3))
        solver.do_numeric_factorization(kkt1)
        dense_check(solver, kkt1, rhs, solver.do_back_solve(rhs))
        # a read-only data array (its address cannot be taken through the buffer protocol: ndarray.ctypes instead)
        ro = np.array(model.block_values(0, 3))
        ro.flags.writeable = False
        kkt1.get_block(0, 0).data = ro
        solver.do_numeric_factorization(kkt1)
        dense_check(solver, kkt1, rhs, solver.do_back_solve(rhs))
        # a block in a float32 / strided .data: not taken by address
        K = kkt1.get_block(1, 1)
        K.data = np.array(model.block_values(1, 3))[::1].astype(np.float32).astype(np.float64)[::-1][::-1]
        solver.do_numeric_factorization(kkt1)
        dense_check(solver, kkt1, rhs, solver.do_back_solve(rhs))
        # every block with index arrays of its own, then rewritten in place into another entry order (first and last
        # entry exchanged): found by the probe of the index arrays
        kkt2 = model.build_kkt(comm=comm, iteration=2)
        for ndx in range(N):
            K = kkt2.get_block(ndx, ndx)
            kkt2.set_block(ndx, ndx, coo_matrix((K.data.copy(), (K.row.copy(), K.col.copy())), shape=K.shape))
        solver.do_numeric_factorization(kkt2)
        solver.do_numeric_factorization(kkt2)
        dense_check(solver, kkt2, rhs, solver.do_back_solve(rhs))
        K = kkt2.get_block(2, 2)
        for a in (K.row, K.col, K.data):
            a[0], a[-1] = a[-1].copy(), a[0].copy()
        solver.do_numeric_factorization(kkt2)
        dense_check(solver, kkt2, rhs, solver.do_back_solve(rhs))
        # values changed in a subset of the entries only (the staging threads compare with what the device already holds and
        # send the column ranges that differ): first a few entries of two blocks, then one whole block, then nothing at all
        kkt_s = model.build_kkt(comm=comm, iteration=2)
        solver.do_numeric_factorization(kkt_s)
        solver.do_numeric_factorization(kkt_s)
        for ndx, pos in ((0, (1, 5)), (3, (0,))):
            K = kkt_s.get_block(ndx, ndx)
            d = K.data.copy()
            diag = np.flatnonzero(K.row == K.col)
            for q in pos:
                d[diag[q]] *= 1.25
            K.data = d
        solver.do_numeric_factorization(kkt_s)
        dense_check(solver, kkt_s, rhs, solver.do_back_solve(rhs))
        kkt_s.get_block(2, 2).data = np.array(model.block_values(2, 4))
        solver.do_numeric_factorization(kkt_s)
        dense_check(solver, kkt_s, rhs, solver.do_back_solve(rhs))
        solver.do_numeric_factorization(kkt_s)
        dense_check(solver, kkt_s, rhs, solver.do_back_solve(rhs))
        # two INTERIOR entries exchanged in place (row, column and value together: the same matrix in another entry
        # order; first, middle and last entry untouched): the checksum of the index arrays finds it -- at once while the
        # arrays fit the per-call byte budget, at the next full check otherwise -- and the block is compared again
        K = kkt2.get_block(3, 3)
        i, j = 2, K.row.size - 3
        assert i != K.row.size >> 1 and j != K.row.size >> 1 and (K.row[i], K.col[i]) != (K.row[j], K.col[j])
        before = dict(eng_calls) if eng_calls is not None else None
        for a in (K.row, K.col, K.data):
            a[i], a[j] = a[j].copy(), a[i].copy()
        solver.do_numeric_factorization(kkt2)
        dense_check(solver, kkt2, rhs, solver.do_back_solve(rhs))
        if eng_calls is not None:
            assert eng_calls['compared_blocks'] > before['compared_blocks']
        # ... and with no byte budget: not later than the `pattern_check_interval`-th call
        solver.pattern_check_bytes, solver.pattern_check_interval = 0, 3
        K = kkt2.get_block(4, 4)
        solver.do_numeric_factorization(kkt2)
        solver._stage_calls = 0
        for a in (K.row, K.col, K.data):
            a[i], a[j] = a[j].copy(), a[i].copy()
        for _ in range(3):
            solver.do_numeric_factorization(kkt2)
        dense_check(solver, kkt2, rhs, solver.do_back_solve(rhs))
        # an interior row index rewritten in place to an entry OUTSIDE the planned pattern: the plan is made again
        # (or an error is raised) -- never a silent solve against the old pattern
        kkt3 = model.build_kkt(comm=comm, iteration=1)
        solver.pattern_check_bytes, solver.pattern_check_interval = 4 << 20, 8
        for ndx in range(N):
            K = kkt3.get_block(ndx, ndx)
            kkt3.set_block(ndx, ndx, coo_matrix((K.data.copy(), (K.row.copy(), K.col.copy())), shape=K.shape))
        solver.do_numeric_factorization(kkt3)
        solver.do_numeric_factorization(kkt3)
        K = kkt3.get_block(1, 1)
        low = np.flatnonzero(K.row > K.col + 1)
        e = int(low[len(low) // 2])
        present = set(zip(K.row.tolist(), K.col.tolist()))
        newrow = next(r for r in range(int(K.col[e]) + 1, K.shape[0]) if (r, int(K.col[e])) not in present)
        up = int(np.flatnonzero((K.row == K.col[e]) & (K.col == K.row[e]))[0])     # its mirror in the upper triangle
        K.row[e] = newrow
        K.col[up] = newrow
        solver.do_numeric_factorization(kkt3)
        dense_check(solver, kkt3, rhs, solver.do_back_solve(rhs))
        # right-hand sides that are not plain float64 vectors go the general way
        rhs2 = model.build_rhs(comm=comm)
        for ndx in range(N):
            rhs2.set_block(ndx, rng.standard_normal(model.block_dim))
        rhs2._blocks[3] = rhs2._blocks[3].astype(np.float32)
        dense_check(solver, kkt3, rhs2, solver.do_back_solve(rhs2))


# ---- zero-pivot test inside a block pivot that mixes scales (interior-point KKT blocks) ------------
def case_mixed_scale_block_pivot(make_engine):
    """A primal variable with a barrier weight of 3e6 next to a constraint row with -1e-7 on its diagonal: the second
    pivot of that pair is -1e-7 - a^2 / 3e6 = -1.7e-7, accurate to all digits.  The zero-pivot test must compare it with
    the terms of ITS OWN sum (pivot.hpp: tmd), not with the largest magnitude of the block pivot it shares with the
    variable (3e6 x 1e-13 > 1.7e-7: rounds 1-2 reported such matrices singular).  A pivot that really cancels to
    rounding noise is still reported."""
    def kkt(d1, c1):
        # x1, x2 | constraint on x1 (diagonal c1), constraint on x1 and x2
        rows = [0, 1, 2, 2, 3, 3, 3]
        cols = [0, 1, 0, 2, 0, 1, 3]
        vals = [d1, 2.0, 0.4654, c1, 0.1, 0.7, 0.0]
        low = coo_matrix((vals, (rows, cols)), shape=(4, 4))
        return (low + sp.tril(low, -1).T).tocoo()
    for d1, c1 in ((3.2e6, -1e-7), (5.4e7, -3.3e-6), (1.0, -1e-7)):
        K = kkt(d1, c1)
        solver = HipLDLInterface(engine=make_engine())
        assert solver.do_symbolic_factorization(K).status == LinearSolverStatus.successful
        # the sequence is fixed on these values; other magnitudes of the same pattern then go through it
        for d1b, c1b in ((d1, c1), (d1 * 1e3, c1), (d1, c1 * 1e-2)):
            Kb = kkt(d1b, c1b)
            res = solver.do_numeric_factorization(Kb, raise_on_error=False)
            assert res.status == LinearSolverStatus.successful, (d1b, c1b)
            assert solver.get_inertia() == (2, 2, 0)
            b = np.array([1.0, -2.0, 0.5, 0.25])
            x = np.asarray(solver.do_back_solve(b))
            assert scaled_residual(Kb, x, b) <= 1e-12
    # genuine cancellation: the Schur complement of the first pivot in the second is 1e-17 of its terms
    low = coo_matrix(([1.0, 1.0, 1.0 + 2.0 ** -52, 3.0], ([0, 1, 1, 2], [0, 0, 1, 2])), shape=(3, 3))
    K = (low + sp.tril(low, -1).T).tocoo()
    solver = HipLDLInterface(engine=make_engine())
    solver.do_symbolic_factorization(K)
    res = solver.do_numeric_factorization(K, raise_on_error=False)
    assert res.status == LinearSolverStatus.singular or solver.get_inertia()[2] == 0    # (a 2x2 pivot may have been chosen)


def case_adversarial_systems(make_engine, seeds=range(0, 400)):
    """tools/fuzz_solver.py --hard through the solver class: zero Hessian entries (2 x 2 pivots) and Jacobian entries scaled
    by up to 1e-7 per instance and iteration -- the adversary of ONE static pivot sequence per pattern group.  Every solution
    the class hands out must have a scaled residual <= 2e-8 against dense algebra (its a-posteriori check, refinement and
    pivot repair: parapint_amd/linalg/solution_check.py; the reference's sub-solvers pivot per block,
    ma27_interface.py:36-47, scipy_interface.py:26-31); it may refuse (RuntimeError) only numerically singular systems.  The
    slice must exercise the machinery: some solves refined, some sequences repaired."""
    import os
    import sys
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools')
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import fuzz_solver
    stats = {}
    bad = [r for r in (fuzz_solver.one(seed, hard=True, engine=make_engine, stats=stats) for seed in seeds) if r is not None]
    assert not bad, bad
    assert stats['solves_refined'] >= 3 and stats['refinement_steps'] >= stats['solves_refined'], stats
    assert stats['inaccurate_solves'] <= 2, stats
    return stats


def case_refinement_fixture(make_engine, device_vectors=False):
    """tests/golden/refinement_case.npz: the pivot sequence of the symbolic phase solves the later matrix with a backward error
    of 0.48; do_back_solve must hand out the refined solution (<= 1e-8 against dense algebra, counted in the solver's
    statistics) -- through host containers, through device vectors, and through do_back_solve_deferred + confirm_solution,
    which must give the very same vector."""
    import os
    from scipy.sparse import coo_matrix
    from parapint_amd.sparse.block_containers import BlockMatrix, BlockVector
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'refinement_case.npz'))
    N = int(d['N'])

    def matrix(tag):
        m = BlockMatrix(N + 1, N + 1)
        for i in range(N + 1):
            for j in range(N + 1):
                key = '%s_%d_%d_val' % (tag, i, j)
                if key in d:
                    pre = '%s_%d_%d_' % (tag, i, j)
                    m.set_block(i, j, coo_matrix((d[pre + 'val'], (d[pre + 'row'], d[pre + 'col'])), shape=tuple(d[pre + 'shape'])))
        return m
    sym, num = matrix('sym'), matrix('num')
    rhs = BlockVector(N + 1)
    for i in range(N + 1):
        rhs.set_block(i, d['rhs_%d' % i])
    Kd = num.toarray()
    Kd = np.tril(Kd) + np.tril(Kd, -1).T
    x_ref = np.linalg.solve(Kd, rhs.flatten())
    solver = new_solver(make_engine, N, result_buffers=2 if device_vectors else 0)
    assert solver.do_symbolic_factorization(sym).status == LinearSolverStatus.successful
    assert solver.do_numeric_factorization(num).status == LinearSolverStatus.successful
    if not device_vectors:
        x = solver.do_back_solve(rhs)
        assert solver.last_residual_first > 1e-3 and solver.last_residual <= 1e-8 and solver.refinement_steps >= 1
        assert scaled_residual(Kd, x.flatten(), rhs.flatten()) <= 1e-8
        assert np.abs(x.flatten() - x_ref).max() <= 1e-6 * np.abs(x_ref).max()
        # unchecked, the same factors hand out the inaccurate solution (what rounds 1-5 did)
        solver.residual_check = False
        x_bad = solver.do_back_solve(rhs)
        assert scaled_residual(Kd, x_bad.flatten(), rhs.flatten()) > 1e-4
        return
    rd = solver.device_vector_from_host(rhs)
    x1 = solver.do_back_solve(rd).to_host(rhs).flatten()
    steps = solver.refinement_steps
    assert steps >= 1 and solver.last_residual <= 1e-8 and scaled_residual(Kd, x1, rhs.flatten()) <= 1e-8
    xd = solver.do_back_solve_deferred(rd)
    assert solver._deferred_solve is not None
    confirmed = solver.confirm_solution()
    assert confirmed is xd and solver.solution_changed_on_confirm and solver.refinement_steps == 2 * steps
    assert np.array_equal(confirmed.to_host(rhs).flatten(), x1)
    # a pending verdict is collected by whatever the caller does next with the solver
    solver.do_back_solve_deferred(rd)
    assert solver.do_numeric_factorization(num).status == LinearSolverStatus.successful and solver._deferred_solve is None
    assert solver.refinement_steps == 3 * steps
