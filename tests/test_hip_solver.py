"""GPU parity suite: the product path (HIP kernels through the C ABI) against the reference's
golden vectors, the oracle and dense algebra.  Run on the GPU box with ``pytest -m gpu``."""
import os

import numpy as np
import pytest

import solver_cases as sc

HERE = os.path.dirname(os.path.abspath(__file__))

pytestmark = pytest.mark.gpu


def make_engine():
    return None          # product default: HipEngine (raises without a GPU / the HIP library)


def test_native_library_is_the_one_in_tree():
    from parapint_amd import _native
    lib = _native.load_library()
    assert _native.LIB_PATH.endswith('parapint_amd/csrc/libparapint_hip.so')
    assert lib is not None


def test_sub_solver_contract(golden):
    sc.case_sub_solver_contract(make_engine, golden)


@pytest.mark.parametrize('mpi', [False, True])
def test_bordered_8x8(golden, mpi):
    sc.case_bordered_8x8(make_engine, golden, mpi)


@pytest.mark.parametrize('mpi', [False, True])
def test_bordered_8x8_with_the_unsymmetric_blocks_of_the_reference_tests(golden, mpi):
    sc.case_bordered_8x8_original(make_engine, golden, mpi)


def test_unsymmetric_blocks_through_the_scipy_interface_route():
    sc.case_general_blocks_random(make_engine)


@pytest.mark.parametrize('shape', [(3, 20, 2, 4), (4, 50, 3, 6)])
def test_small_synthetic(golden, shape):
    sc.case_small_synthetic(make_engine, golden, shape)


def test_known_answer(golden):
    sc.case_known_answer(make_engine, golden)


def test_oracle_schur_and_solution():
    sc.case_oracle_schur(make_engine, (5, 40, 2, 8))
    sc.case_oracle_schur(make_engine, (70, 30, 2, 10))     # more than one 64-instance wave, ragged tail


def test_heterogeneous_groups():
    sc.case_heterogeneous(make_engine)


def test_error_behaviour():
    sc.case_errors(make_engine)


def test_singular_schur_complement_is_a_status_when_asked_for_one():
    sc.case_singular_schur_complement(make_engine)


def test_nested_block_matrices_and_vectors():
    sc.case_nested_blocks(make_engine)


def test_sparse_corner_entry_point_checks_its_arguments():
    """pp_factor_schur_corner is the block-tridiagonal form of pp_factor_schur: status 3 on a dense S and for positions
    outside the Schur buffer; the flat and the sparse form of Q give the same factorisation."""
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.results import LinearSolverStatus
    solver, model = sc.case_dynamic(make_engine, 2, 3, expect_block_tridiagonal=False, oracle=False)
    with pytest.raises(RuntimeError):
        solver._eng.factor_schur_corner(np.array([0], dtype=np.int64), np.array([1.0]))
    solver, model = sc.case_dynamic(make_engine, 12, 5, expect_block_tridiagonal=True, oracle=False, dense_limit=8)
    gs, G = solver._btd
    with pytest.raises(RuntimeError):
        solver._eng.factor_schur_corner(np.array([(2 * G - 1) * gs * gs], dtype=np.int64), np.array([1.0]))
    kkt = model.build_kkt(comm=SerialComm(), iteration=2)
    rhs = model.build_rhs(comm=SerialComm())
    assert solver.do_numeric_factorization(kkt).status == LinearSolverStatus.successful
    x1 = solver.do_back_solve(rhs).flatten()
    inertia = solver.get_inertia()
    solver._eng.factor_schur_flat(solver._btd_q(solver._last_Q))           # the flat form on the same all-reduced S
    assert solver._eng.status()[0] == 0
    x2 = solver.do_back_solve(rhs).flatten()
    assert np.abs(x1 - x2).max() <= 1e-12 * np.abs(x1).max()
    assert solver.get_inertia() == inertia


def test_inertia_correction_pattern_growth():
    sc.case_inertia_correction_pattern_growth(make_engine)


def test_pivot_order_refresh_after_static_breakdown():
    sc.case_pivot_order_refresh(make_engine)


def test_instances_of_one_group_that_need_different_pivot_sequences():
    """Round 5: the group is split into variants instead of reporting a regular matrix singular (ma27_interface.py:110-140)."""
    sc.case_conflicting_pivots(make_engine)


def test_ip_solve_call_pattern():
    sc.case_ip_solve_call_pattern(make_engine)


def test_memory_reallocation_retry_loop():
    """Device-storage budget too small -> status not_enough_memory from the numeric phase -> the caller's
    increase_memory_allocation loop -> success (linalg/tests/test_realloc.py:10-61, interior_point.py:634-652)."""
    need = sc.case_reallocation(make_engine, lambda solver: solver._eng.memory_info()[0])
    assert need > 1000


def test_reference_realloc_matrix_through_the_sub_solver_adapters():
    """The reference's own reallocation fixture (linalg/tests/test_realloc.py:10-61: 10 000 rows, 2x2 pivots everywhere)
    through MumpsInterface / InteriorPointMA27Interface / ScipyInterface on the device."""
    sc.case_reference_realloc_matrix(make_engine, n=10000)


def test_back_solve_results_do_not_alias():
    """Default result_buffers = 0: every do_back_solve returns storage of its own (host and device vectors), as the
    reference does; the rotating pool is an opt-in."""
    from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
    from parapint_amd.linalg.comm import SerialComm
    model = SyntheticKKT(6, 30, 2, 5)
    solver = sc.new_solver(make_engine, 6)
    kkt = model.build_kkt(comm=SerialComm(), iteration=1)
    rhs = model.build_rhs(comm=SerialComm())
    solver.do_symbolic_factorization(kkt)
    solver.do_numeric_factorization(kkt)
    xs = [solver.do_back_solve(rhs) for _ in range(4)]
    keep = xs[0].flatten().copy()
    rhs2 = model.build_rhs(comm=SerialComm())
    for ndx in range(6):
        rhs2.set_block(ndx, 2.0 * np.asarray(rhs2.get_block(ndx)))
    for _ in range(3):
        solver.do_back_solve(rhs2)
    assert np.array_equal(xs[0].flatten(), keep)
    rd = solver.device_vector_from_host(rhs)
    d0 = solver.do_back_solve(rd)
    ref = d0.group_tensors[0].clone()
    rd2 = solver.device_vector_from_host(rhs2)
    for _ in range(3):
        d1 = solver.do_back_solve(rd2)
    assert d1.group_tensors[0].data_ptr() != d0.group_tensors[0].data_ptr()
    assert bool((d0.group_tensors[0] == ref).all())


@pytest.mark.parametrize('shape', [(1, 10, 2, 1), (1, 5, 2, 5), (65, 10, 2, 2), (129, 12, 2, 3), (5, 300, 2, 208),
                                   (3, 300, 2, 209), (2, 600, 2, 513), (2, 800, 2, 730), (2, 1000, 2, 1000),
                                   (2, 1001, 2, 1001)])
def test_edge_shapes_against_full_space_superlu(shape):
    """One block, one coupling variable, n_theta = n_q, a ragged chunk holding a single instance, and the coupling
    dimensions at which the dense S path changes (208 | 209: register-resident / blocked, 512 | 513: one workgroup /
    multi-workgroup factorisation and the wide coupling solve)."""
    sc.case_against_oracle(make_engine, shape, iteration=1)


def test_config2_against_full_space_superlu():
    # BASELINE.json configs[1]: 64 scenarios x 2k primal vars/block, 100 coupling vars
    solver, model = sc.case_against_oracle(make_engine, (64, 400, 4, 100), iteration=1)
    st = solver.plan_stats[0]
    assert st['n'] == 3700 and st['batch'] == 64


def test_refactor_with_new_values_same_object():
    from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
    from parapint_amd.linalg.comm import SerialComm
    model = SyntheticKKT(10, 50, 3, 10)
    solver = sc.new_solver(make_engine, 10)
    rhs = model.build_rhs(comm=SerialComm())
    solver.do_symbolic_factorization(model.build_kkt(comm=SerialComm()))
    for it in (0, 1, 2):
        kkt = model.build_kkt(comm=SerialComm(), iteration=it)
        solver.do_numeric_factorization(kkt)
        x = solver.do_back_solve(rhs)
        assert sc.scaled_residual(kkt.tocoo(), x.flatten(), rhs.flatten()) <= sc.RESID_TOL


def test_full_size_blocks_property():
    # configuration-3 block shape (n_i = 9200, n_c = 200) on 130 instances: residual + inertia properties
    solver, model = sc.case_against_oracle(make_engine, (130, 1000, 4, 200), iteration=4, check_full_space=False)
    assert solver.plan_stats[0]['n'] == 9200


def test_full_size_c3_residual_inertia_determinism():
    """BASELINE.json configs[2] at its full size through the LinearSolverInterface boundary: 1024 blocks x 9200,
    200 coupling variables; two value sets, scaled residual <= 1e-8 on the assembled system, inertia, bitwise
    determinism of a repeated factorisation + solve."""
    from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
    from parapint_amd.linalg.comm import SerialComm
    N = 1024
    model = SyntheticKKT(N, 1000, 4, 200)
    solver = sc.new_solver(make_engine, N)
    rhs = model.build_rhs(comm=SerialComm())
    b = rhs.flatten()
    want = (N * (model.n_y + 1000) + 200, N * (model.n_y + 200), 0)
    for it in (0, 3):
        kkt = model.build_kkt(comm=SerialComm(), iteration=it)
        if it == 0:
            solver.do_symbolic_factorization(kkt)
        xs = []
        for rep in range(2):
            assert solver.do_numeric_factorization(kkt).status.value == 0
            xs.append(solver.do_back_solve(rhs).flatten())
        assert np.array_equal(xs[0], xs[1])
        assert sc.scaled_residual(kkt.tocoo(), xs[0], b) <= sc.RESID_TOL
        assert tuple(solver.get_inertia()) == want
    st = solver.plan_stats[0]
    assert st['n'] == 9200 and st['batch'] == 1024


def test_config5_shaped_blocks():
    """BASELINE.json configs[4] block shape (n_q = 2000, m = 4: 10 000 primal variables, block dimension 19 000,
    n_c = 1000: the multi-workgroup dense LDL^T and the 1024-thread coupling solve) on 72 blocks: S on a subset of its
    columns against the reference algorithm restated with SuperLU (one solve per border row, mpi_...:313-333),
    residual, inertia."""
    from scipy.sparse.linalg import splu
    from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
    from parapint_amd.linalg.comm import SerialComm
    N, n_q, m, n_t = 72, 2000, 4, 1000
    model = SyntheticKKT(N, n_q, m, n_t)
    kkt = model.build_kkt(comm=SerialComm(), iteration=2)
    rhs = model.build_rhs(comm=SerialComm())
    solver = sc.new_solver(make_engine, N)
    solver.do_symbolic_factorization(kkt)
    assert solver.do_numeric_factorization(kkt).status.value == 0
    x = solver.do_back_solve(rhs)
    assert solver.plan_stats[0]['n'] == 19000
    assert sc.scaled_residual(kkt.tocoo(), x.flatten(), rhs.flatten()) <= sc.RESID_TOL
    assert tuple(solver.get_inertia()) == (N * (model.n_y + n_q) + n_t, N * (model.n_y + n_t), 0)
    cols = [0, 1, 137, 500, 998, 999]
    A = model.border_matrix().tocsr()
    S_ref = np.zeros((n_t, len(cols)))
    for ndx in range(N):
        lu = splu(model.block_matrix(ndx, 2).tocsc())
        for j, r in enumerate(cols):
            S_ref[:, j] -= A @ lu.solve(A[r].toarray().ravel())
    S = solver.get_schur_complement()
    assert np.abs(S[:, cols] - S_ref).max() <= 1e-9 * np.abs(S_ref).max()


def test_device_resident_matrix_and_vectors_f2():
    """the KKT values gathered on the device from the interface's own arrays (Hessian diagonal, Jacobian
    values; value map of sc_ip_interface.py:1677-1681 / interface.py:432-494) and device-resident right-hand sides /
    solutions, against the host COO path of the same class on the same values: S and x bit for bit (the kernels see
    identical inputs), and against the assembled system by residual.  70 blocks = a full wave + a ragged one."""
    import torch
    from scipy.sparse import coo_matrix
    from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.sparse.block_containers import BlockMatrix
    N = 70
    model = SyntheticKKT(N, 30, 2, 10)
    rhs = model.build_rhs(comm=SerialComm())
    dev = sc.new_solver(make_engine, N)
    dk = model.build_device_kkt(comm=SerialComm())
    assert dev.do_symbolic_factorization(matrix=dk, raise_on_error=False).status.value == 0
    host = sc.new_solver(make_engine, N)
    host.do_symbolic_factorization(model.build_kkt(comm=SerialComm(), iteration=0))
    rhs_dev = dev.device_vector_from_host(rhs)
    base_sources = dict(dk.sources)
    for it, per_entry in ((3, False), (4, True)):
        srcs = {ndx: model.block_sources(ndx, it, per_entry=per_entry) for ndx in range(N)}
        if per_entry:                                   # a second value set in tensors of its own (what a producer cycles)
            dk_it = dk.with_sources({gid: torch.zeros_like(t) for gid, t in base_sources.items()})
        else:
            dk_it = dk
        dk_it.set_sources_from_host(srcs)
        assert dev.do_numeric_factorization(matrix=dk_it, raise_on_error=False).status.value == 0
        x = dev.do_back_solve(rhs_dev)
        assert x.get_block(0).is_cuda and x.get_block(N).is_cuda
        xh = x.to_host(rhs)
        # the same values through the host boundary
        kkt = BlockMatrix(N + 1, N + 1)
        A = model.border_matrix()
        for ndx in range(N):
            kv, bv = model.block_values_from_sources(srcs[ndx])
            kkt.set_block(ndx, ndx, coo_matrix((kv, (model._row, model._col)), shape=(model.block_dim,) * 2))
            Ai = coo_matrix((bv, (A.row, A.col)), shape=A.shape)
            kkt.set_block(N, ndx, Ai)
            kkt.set_block(ndx, N, Ai.transpose().tocoo())          # (only read by the residual check below)
        kkt.set_block(N, N, coo_matrix((10, 10)))
        host.do_numeric_factorization(kkt)
        x_ref = host.do_back_solve(rhs)
        assert np.array_equal(dev.get_schur_complement(), host.get_schur_complement())
        assert np.array_equal(xh.flatten(), x_ref.flatten())
        assert dev.get_inertia() == host.get_inertia()
        assert sc.scaled_residual(kkt.tocoo(), xh.flatten(), rhs.flatten()) <= sc.RESID_TOL
    # a host matrix can still be given to the same solver object afterwards
    dev.do_numeric_factorization(kkt)
    assert np.array_equal(dev.do_back_solve(rhs).flatten(), x_ref.flatten())


def test_dense_schur_paths_agree():
    """S of the synthetic KKT is positive definite: the blocked MFMA LDL^T must be accepted and
    give the same solution as the Bunch-Kaufman kernel (ragged last panel: n_c = 100, 200, 37)."""
    from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
    from parapint_amd.linalg.comm import SerialComm
    for shape in [(8, 120, 2, 37), (6, 400, 2, 100), (4, 1000, 2, 200)]:
        model = SyntheticKKT(*shape)
        kkt = model.build_kkt(comm=SerialComm(), iteration=1)
        rhs = model.build_rhs(comm=SerialComm())
        sols, modes = [], []
        for policy in (0, 1):
            solver = sc.new_solver(make_engine, shape[0])
            solver._eng.set_dense_policy(policy)
            solver.do_symbolic_factorization(kkt)
            solver.do_numeric_factorization(kkt)
            modes.append(solver._eng.dense_mode())
            x = solver.do_back_solve(rhs)
            assert sc.scaled_residual(kkt.tocoo(), x.flatten(), rhs.flatten()) <= sc.RESID_TOL
            sols.append((x.flatten(), solver.get_inertia()))
        assert modes == [1, 0]
        assert sols[0][1] == sols[1][1]
        assert np.abs(sols[0][0] - sols[1][0]).max() <= 1e-9 * np.abs(sols[1][0]).max()


def test_indefinite_schur_falls_back_to_bunch_kaufman():
    # heterogeneous case has an indefinite S + Q: the optimistic factor must be rejected on device
    import numpy as np
    from scipy.sparse import coo_matrix
    from parapint_amd.sparse.block_containers import BlockMatrix, BlockVector
    rng = np.random.default_rng(3)
    nb, nc = 3, 40
    A = BlockMatrix(nb + 1, nb + 1)
    rhs = BlockVector(nb + 1)
    full = np.zeros((nb * 30 + nc, nb * 30 + nc))
    for i in range(nb):
        M = rng.normal(size=(30, 30))
        K = M @ M.T + 30 * np.eye(30)
        B = rng.normal(size=(nc, 30))
        A.set_block(i, i, coo_matrix(K))
        A.set_block(nb, i, coo_matrix(B))
        rhs.set_block(i, rng.normal(size=30))
        full[30 * i:30 * i + 30, 30 * i:30 * i + 30] = K
        full[nb * 30:, 30 * i:30 * i + 30] = B
        full[30 * i:30 * i + 30, nb * 30:] = B.T
    Q = rng.normal(size=(nc, nc))
    Q = Q + Q.T                                   # indefinite coupling block
    A.set_block(nb, nb, coo_matrix(Q))
    full[nb * 30:, nb * 30:] = Q
    rhs.set_block(nb, rng.normal(size=nc))
    solver = sc.new_solver(make_engine, nb)
    solver.do_symbolic_factorization(A)
    solver.do_numeric_factorization(A)
    assert solver._eng.dense_mode() == 0
    x = solver.do_back_solve(rhs)
    x_ref = np.linalg.solve(full, rhs.flatten())
    assert np.abs(x.flatten() - x_ref).max() <= 1e-8 * np.abs(x_ref).max()
    ev = np.linalg.eigvalsh(full)
    assert solver.get_inertia() == (int((ev > 0).sum()), int((ev < 0).sum()), 0)


@pytest.mark.parametrize('wmax,tol', [(2, 1), (4, 1), (4, 3)])
def test_block_pivots_on_device(wmax, tol):
    """Supernodes (merged sub-pivot chains, block pivots up to 4 wide) against the oracle and dense algebra."""
    def make():
        from parapint_amd.linalg.hip_schur_complement import HipEngine
        eng = HipEngine()
        eng.set_supernodes(wmax, tol)
        return eng
    sc.case_oracle_schur(make, (70, 30, 2, 10))
    sc.case_heterogeneous(make)
    solver, model = sc.case_against_oracle(make, (16, 400, 4, 100), iteration=2)
    assert solver.plan_stats[0]['n_levels'] < 30


@pytest.mark.parametrize('wmax', [1, 4])
def test_device_factor_matches_host_interpreter(wmax):
    """The factor storage itself (unscaled panels U, scaled rows L = the MA27 factor entries, block-pivot
    inverses) of every block, read back through pp_get_factor, against the one-instance host interpreter of
    the same plan (tests/hostsim) -- catches any device-side miscompile or indexing slip that the solution
    alone could mask.  70 blocks = one full 64-instance wave + a ragged one."""
    from hostsim_util import HostSim, lib as hostlib
    from parapint_amd.linalg.hip_schur_complement import HipEngine
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
    N = 70
    model = SyntheticKKT(N, 30, 2, 10)
    kkt = model.build_kkt(comm=SerialComm(), iteration=2)
    eng = HipEngine()
    eng.set_supernodes(wmax, 1)
    solver = sc.new_solver(lambda: eng, N)
    solver.do_symbolic_factorization(kkt)
    solver.do_numeric_factorization(kkt)
    hostlib().ppsim_set_supernodes(wmax, 1)
    try:
        A = model.border_matrix()
        hs = HostSim(model.block_matrix(0, 2), A)          # the group's plan is made on its first block
        st = eng.ns.group_stats(0)
        assert st['u_doubles'] == hs.stats['usize'] and st['n_pivots'] == hs.stats['npiv']
        for inst in (0, 1, 37, 63, 64, 69):
            rc, _, _ = hs.factor(hs.canonical(model.block_matrix(inst, 2), A))
            assert rc == 0
            for which, ref in ((0, hs.U), (1, hs.L), (2, hs.Dinv)):
                dev = eng.get_factor(0, which, inst, ref.size)
                assert np.abs(dev - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()), (inst, which)
    finally:
        hostlib().ppsim_set_supernodes(0, -1)


@pytest.mark.parametrize('shape', [(72, 16, 2, 8), (12, 49, 2, 6)])
def test_chain_front_factor_matches_host_interpreter(shape):
    """Round 5: the panels of a chain front (plan.hpp) are factorised by k_chain_front -- one workgroup per instance, the
    front's pivot columns in LDS -- into the SAME U / L / pivot-inverse storage the level kernels fill.  Every instance of
    the interior time blocks (one full wave + a ragged one / a few) against the host interpreter of the same plan, which
    mirrors the front kernel's sums; S, solution and inertia of the whole problem against the oracle besides."""
    import ctypes
    from hostsim_util import HostSim, lib as hostlib, _ip
    T, n_s, n_u, nfe = shape
    solver, model = sc.case_dynamic(make_engine, T, n_s, n_u=n_u, nfe=nfe)
    eng = solver._eng
    g = max(solver._groups, key=lambda q: len(q.blocks))
    assert len(g.blocks) == T - 2
    L = hostlib()
    L.ppsim_set_batch_hint(len(g.blocks))
    L.ppsim_set_mapped_hint(1)
    try:
        def block(t, it):
            A = model.border_matrix(t).tocsr()
            return model.block_matrix(t, it).tocoo(), A[np.unique(A.tocoo().row), :]
        K0, A0 = block(g.blocks[0], 1)                   # (the plan is made on the group's first block at the symbolic phase)
        hs = HostSim(K0, A0)
        st = eng.ns.group_stats(g.gid)
        assert st['u_doubles'] == hs.stats['usize'] and st['n_pivots'] == hs.stats['npiv']
        cs = np.zeros(5, dtype=np.int32)
        L.ppsim_chain_stats(hs.h, _ip(cs))
        assert cs[0] >= 1 and cs[2] < hs.stats['n_levels']          # chain fronts exist and shorten the factor schedule
        for slot in sorted(set((0, 1, len(g.blocks) // 2, len(g.blocks) - 1))):
            K, A = block(g.blocks[slot], 2)               # (case_dynamic's last factorisation: iteration + 1)
            rc, _, _ = hs.factor(hs.canonical(K, A))
            assert rc == 0
            for which, ref in ((0, hs.U), (1, hs.L), (2, hs.Dinv)):
                dev = eng.get_factor(g.gid, which, slot, ref.size)
                assert np.abs(dev - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max()), (slot, which)
    finally:
        L.ppsim_set_batch_hint(0)
        L.ppsim_set_mapped_hint(0)


@pytest.mark.parametrize('shape', [(3, 220, 2, 220), (2, 530, 2, 530)])
def test_large_coupling_dimension_paths(shape):
    """n_c above the register-resident dense factor (208) and above one-thread-per-unknown solve (512):
    the blocked global-memory LDL^T and the blocked solve must give the same answers."""
    sc.case_oracle_schur(make_engine, shape)


def test_inertia_correction_fast_path_on_device():
    """one retry of the inertia-correction loop from resident values -- refactorize_with_diagonal_shift
    (+delta on the Hessian diagonals, -delta on the constraint diagonals, +delta on the coupling block; applied on the
    device, no staging / H2D) against a fresh numeric factorisation of the regularised matrix built on the host the way
    interfaces/interface.py:590-619 and sc_ip_interface.py:1736-1757 build it."""
    import scipy.sparse as sp
    from scipy.sparse import coo_matrix
    from parapint_amd.sparse.block_containers import BlockMatrix, BlockVector
    from parapint_amd.linalg.results import LinearSolverStatus
    rng = np.random.default_rng(21)
    n_x, n_c, nc, nb = 12, 5, 4, 70                  # 70 blocks: a full wave + a ragged one
    H, J, B = [], [], []
    h0 = rng.uniform(0.5, 2.0, size=n_x)
    Jp = (sp.random(n_c, n_x, density=0.3, random_state=3, data_rvs=lambda k: rng.normal(size=k)) + 2.0 * sp.eye(n_c, n_x)).tocoo()
    Bp = coo_matrix((np.ones(nc), (np.arange(nc), rng.choice(n_x, nc, replace=False))), shape=(nc, n_x + n_c))
    for i in range(nb):
        H.append(sp.diags(h0 * rng.uniform(0.8, 1.2, size=n_x)).tocoo())
        J.append(coo_matrix((Jp.data * rng.uniform(0.8, 1.2, size=Jp.nnz), (Jp.row, Jp.col)), shape=Jp.shape))
        B.append(coo_matrix((rng.normal(size=nc), (Bp.row, Bp.col)), shape=Bp.shape))

    def kkt(dw, dc):
        A = BlockMatrix(nb + 1, nb + 1)
        for i in range(nb):
            Hi = (H[i] + dw * sp.identity(n_x, format='coo')).tocoo()
            Ci = (-dc) * sp.identity(n_c, format='coo')            # explicit (possibly zero) constraint diagonal
            A.set_block(i, i, sp.bmat([[Hi, J[i].T], [J[i], Ci]]).tocoo())
            A.set_block(nb, i, B[i])
        A.set_block(nb, nb, (dw * sp.identity(nc, format='coo')).tocoo())
        return A

    rhs = BlockVector(nb + 1)
    for i in range(nb):
        rhs.set_block(i, rng.normal(size=n_x + n_c))
    rhs.set_block(nb, rng.normal(size=nc))
    classes = {i: np.concatenate([np.ones(n_x, dtype=np.int8), 2 * np.ones(n_c, dtype=np.int8)]) for i in range(nb)}

    from oracle.schur_complement import SchurComplementLinearSolver as OracleSC     # serial twin: a BlockMatrix
    from oracle.subsolvers import ScipyInterface as OracleScipy
    fast = sc.new_solver(make_engine, nb)
    A0 = kkt(0.0, 0.0)
    fast.do_symbolic_factorization(A0)
    fast.do_numeric_factorization(A0)
    fast.set_regularization_classes(classes)
    m = n_x + n_c
    for dw, dc in ((1e-4, 1e-4), (1e-2, 1e-2), (1.0, 1e-8)):
        res = fast.refactorize_with_diagonal_shift(dw, dc, coupling_shift=dw, raise_on_error=False)
        assert res.status == LinearSolverStatus.successful
        # the regularised matrix assembled the reference's way, solved by the oracle's restatement of the reference
        # algorithm (n_c solves per block, SuperLU sub-solver) and by dense algebra
        Areg = kkt(dw, dc)
        oracle = OracleSC({i: OracleScipy() for i in range(nb)}, OracleScipy())
        oracle.do_symbolic_factorization(Areg)
        assert oracle.do_numeric_factorization(Areg).status.value == 0
        S_o = oracle.schur_complement - dw * np.eye(nc)      # the serial class starts S from Q (explicit_...:108); ours is without
        S1 = fast.get_schur_complement()
        assert np.abs(S1 - S_o).max() <= 1e-9 * max(1.0, np.abs(S_o).max())
        x_o = oracle.do_back_solve(rhs.copy()).flatten()     # (quirk Q4: the serial class updates its rhs in place)
        full = np.zeros((nb * m + nc, nb * m + nc))
        for i in range(nb):
            full[i * m:(i + 1) * m, i * m:(i + 1) * m] = Areg.get_block(i, i).toarray()
            Bd = Areg.get_block(nb, i).toarray()
            full[nb * m:, i * m:(i + 1) * m] = Bd
            full[i * m:(i + 1) * m, nb * m:] = Bd.T
        full[nb * m:, nb * m:] = dw * np.eye(nc)
        ev = np.linalg.eigvalsh(full)
        assert fast.get_inertia() == (int((ev > 0).sum()), int((ev < 0).sum()), 0)
        x_d = np.linalg.solve(full, rhs.flatten())
        x1 = fast.do_back_solve(rhs).flatten()
        assert np.abs(x1 - x_o).max() <= 1e-8 * np.abs(x_o).max()
        assert np.abs(x1 - x_d).max() <= 1e-8 * np.abs(x_d).max()
    # the classes survive a new plan (pivot-order refresh / union pattern): the fast path keeps working
    fast._run_symbolic()
    fast.do_numeric_factorization(A0)
    assert fast.refactorize_with_diagonal_shift(1e-3, 1e-3, coupling_shift=1e-3).status == LinearSolverStatus.successful
    # a classed row without a diagonal entry in the plan is refused with a message, not silently skipped
    A1 = BlockMatrix(nb + 1, nb + 1)
    for i in range(nb):
        A1.set_block(i, i, sp.bmat([[H[i], J[i].T], [J[i], None]]).tocoo())
        A1.set_block(nb, i, B[i])
    A1.set_block(nb, nb, coo_matrix((nc, nc)))
    bare = sc.new_solver(make_engine, nb)
    bare.do_symbolic_factorization(A1)
    bare.do_numeric_factorization(A1)
    with pytest.raises(RuntimeError, match='no diagonal entry'):
        bare.set_regularization_classes(classes)


@pytest.mark.parametrize('method', ['ssc', 'fs'])
def test_performance_harness_known_answer(method, capsys):
    """The harness counterpart of examples/performance/schur_complement/main.py on the reference's known-answer
    configuration (examples/tests/test_examples.py:76-99: 0.3163456780448639)."""
    from parapint_amd.examples.performance.schur_complement import main as harness
    res = harness.run(harness.parse_args(['--method', method, '--n_blocks', '3', '--n_q_per_block', '500',
                                          '--n_y_multiplier', '12']))
    assert abs(res.max_err - 0.3163456780448639) <= 5e-8
    out = capsys.readouterr().out
    assert 'Est Err' in out and 'Num Fact (s)' in out and ('%.10f' % res.max_err) in out


def test_rccl_collectives_on_solver_buffers():
    """The two data-path all-reduces through RCCL on the solver's own device buffers and stream (one-rank group:
    a one-GPU box cannot host two RCCL ranks; the two-rank host logic is covered by test_multirank_gloo.py)."""
    import os
    import socket
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(here, 'rccl_worker.py')]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = out.stdout.decode()
    assert out.returncode == 0, text[-4000:]
    assert 'rccl one-rank ok' in text


def _run_bench(extra_env, *flags):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update(extra_env)
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + list(flags), env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=900)
    text = out.stdout.decode()
    assert out.returncode == 0, (text[-2000:], out.stderr.decode()[-4000:])
    line = [ln for ln in text.splitlines() if ln.startswith('{')][-1]
    return json.loads(line)


def test_bench_two_ranks_started_by_the_script_itself():
    """`python bench.py --gpus 2` starts its own ranks.  Here the two ranks share ONE device and exchange through gloo
    (PP_BENCH_REHEARSAL: rehearsal of the N > 1 path -- ownership, the packed all-reduce with the status tail, max-over-ranks
    timing, the correctness gate inside the bench); the run with one GPU per rank over RCCL is tests/test_zz_two_devices.py
    (last in the order, skipped on a one-GPU box)."""
    import torch
    res = _run_bench({'PP_BENCH_REHEARSAL': 'gloo'}, '--gpus', '2', '--workload', 'C2', '--steps', '4',
                     '--warmup', '2', '--no-cpu-baseline', '--boundary-iterations', '2', '--profile-steps', '1')
    assert res['n_gpus'] == 2 and res['correct'] is True
    assert res['config']['world_size'] == 2 and res['config']['blocks_per_gpu'] == 32
    assert res['config']['collective_backend'] == 'gloo'
    assert res['residual'] <= 1e-8 and res['inertia'] == res['expected_inertia']
    assert len(res['collective_us']['allreduce_S_and_status']) == 2 and min(res['collective_us']['allreduce_r_s']) > 0.0
    assert res['scaling'] == 'strong'
    # the collectives of one step: [S | status], r_s, and the agreement of the a-posteriori check (coupling sums + one slot per rank)
    assert res['solution_check']['on'] is True and res['solution_check']['collectives_per_step'] == 3
    assert res['solution_check']['backward_error_last_step'] <= 1e-10
    if torch.cuda.device_count() < 2:
        # one process per GPU: more ranks than devices is refused with a clear message (not inside ncclCommInitRank)
        import os
        import subprocess
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = {k: v for k, v in os.environ.items() if k != 'PP_BENCH_REHEARSAL'}
        out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--workload', 'C2', '--steps', '2',
                              '--no-cpu-baseline', '--no-boundary', '--no-ip-loop'], env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, timeout=600)
        assert out.returncode != 0 and 'HIP device(s) visible' in out.stderr.decode()


def test_bench_single_rank_line_has_the_contract_fields():
    res = _run_bench({}, '--workload', 'C2', '--steps', '5', '--warmup', '2', '--no-cpu-baseline', '--boundary-iterations', '2')
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'boundary_host', 'device_only',
                'value_no_prefetch', 'ms_per_step_no_prefetch', 'value_boundary_constant_declared',
                'value_boundary_flat_values', 'value_unchecked', 'solution_check', 'shares'):
        assert key in res
    # the timed steps are CHECKED steps (every back-solve ends with the residual on the device); the unchecked rate beside it
    assert res['solution_check']['on'] is True and res['solution_check']['backward_error_last_step'] <= 1e-10
    assert res['value_unchecked'] > 0          # (normally the faster of the two; a rate of five steps is not asserted against another)
    assert res['value_no_prefetch'] > 0 and res['value_boundary_constant_declared'] > 0
    assert res['boundary_host_constant_declared']['residual'] <= 1e-8
    assert res['boundary_host_flat_values']['residual'] <= 1e-8 and res['boundary_host_flat_values']['constant_declared']['residual'] <= 1e-8
    assert res['correct'] is True and res['n_gpus'] == 1 and res['roofline']['bound'] == 'hbm'
    assert abs(res['value'] * res['ms_per_step'] / 1e3 - 1.0) < 1e-6


def test_bench_c4_full_size_dynamic_workload():
    """BASELINE.json configs[3] at full size through the bench's own gate: 512 time blocks x ~4k variables, n_s = 49,
    block-tridiagonal S of dimension 50 078 by cyclic reduction (residual <= 1e-8, inertia exact)."""
    res = _run_bench({}, '--workload', 'C4', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-boundary',
                     '--profile-steps', '1')
    assert res['correct'] is True and res['residual'] <= 1e-8 and res['inertia'] == res['expected_inertia']
    assert res['config']['blocks_per_gpu'] == 512


def test_bench_c5_one_rank_share():
    """BASELINE.json configs[4], the share of one rank at 8 GPUs: 512 scenarios x 19 000 rows, 1000 coupling variables
    (dense 1000 x 1000 S on the matrix cores), through the bench's gate."""
    res = _run_bench({}, '--workload', 'C5', '--blocks', '512', '--steps', '3', '--warmup', '1', '--no-cpu-baseline',
                     '--no-boundary', '--profile-steps', '1')
    assert res['correct'] is True and res['residual'] <= 1e-8 and res['inertia'] == res['expected_inertia']
    assert res['plan']['n'] == 19000 and res['config']['blocks_per_gpu'] == 512


def test_bench_line_reports_the_ip_loop_and_the_boundary_rate():
    """Round 3: the default line carries `ip_loop` (the interior-point loop with device-resident iterates converged) and
    `value_boundary` (SURVEY 8(d) to the letter: host containers in and out)."""
    res = _run_bench({}, '--steps', '5', '--warmup', '2', '--no-cpu-baseline', '--boundary-iterations', '2',
                     '--ip-scenarios', '256', '--profile-steps', '1', '--ip-time-blocks', '64')
    assert res['correct'] is True
    ipl = res['ip_loop']
    assert ipl['converged'] is True and ipl['iterations'] > 5 and ipl['it_per_s'] > 0
    # round 4: the C3-shaped QP (5000 primal variables with bounds per scenario, 200 first-stage variables), every step of
    # the loop a kernel of the library (no torch operator inside the iterations), step kernels timed against the HBM roof
    assert ipl['primal_variables_per_scenario'] == 5000 and ipl['block_dim'] == 9200 and ipl['n_coupling'] == 200
    assert ipl['scenarios'] == 256 and max(ipl['final_infeasibilities']) <= 1e-8 and ipl['torch_ops_per_iteration'] == 0
    assert ipl['pivot_order_refreshes'] == 0
    assert set(ipl['step_kernels']) == {'rhs', 'step_lengths', 'take_step', 'residuals'}
    assert all(0.0 < v['frac_of_hbm_peak'] < 1.0 for v in ipl['step_kernels'].values())       # (reported, not gated: a rate is not a parity property)
    assert res['value_boundary'] == res['boundary_host']['it_per_s'] > 0
    # the time-staged counterpart (here 64 time blocks of the C4 shape): converged, block-tridiagonal coupling block
    dyn = res['ip_loop_dynamic']
    assert dyn['converged'] is True and dyn['time_blocks'] == 64 and dyn['n_coupling'] == 2 * 49 * 63
    assert dyn['block_dim'] == 4254 and max(dyn['final_infeasibilities']) <= 1e-8 and dyn['torch_ops_in_the_loop'] <= 6
    # and BASELINE configs[3] itself (the nonlinear Burgers discretisation; here 64 of its 512 time blocks)
    bur = res['ip_loop_burgers']
    assert bur['converged'] is True and bur['variables_per_block'] == 4018 and bur['n_coupling'] == 2 * 49 * 63
    assert max(bur['final_infeasibilities']) <= 1e-8 and bur['iterations'] <= 8


def test_pinned_host_memory_is_bounded():
    """pp_host_alloc hands out at most PP_PINNED_LIMIT_MB of page-locked memory per process (NULL beyond: the engine's
    pageable fallback) and gives freed memory back to the budget."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import ctypes, numpy as np\n"
        "from parapint_amd.linalg.hip_engine import HipEngine\n"
        "e = HipEngine(); lib = e.lib\n"
        "a = lib.pp_host_alloc(ctypes.c_int64(3 << 20)); assert a\n"
        "b = lib.pp_host_alloc(ctypes.c_int64(2 << 20)); assert not b          # 3 + 2 MiB > 4 MiB\n"
        "lib.pp_host_free(a)\n"
        "c = lib.pp_host_alloc(ctypes.c_int64(4 << 20)); assert c; lib.pp_host_free(c)\n"
        "big = e.alloc_pinned((1 << 20,)); assert big.shape == (1 << 20,) and float(big.sum()) == 0.0   # 8 MiB: pageable fallback\n"
        "print('pinned cap ok')\n") % root
    env = dict(os.environ, PP_PINNED_LIMIT_MB='4')
    out = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert out.returncode == 0 and 'pinned cap ok' in out.stdout.decode(), out.stdout.decode()[-2000:]


def test_pivot_growth_guard():
    sc.case_growth_guard(make_engine)


@pytest.mark.parametrize('device_vectors', [False, True])
def test_inaccurate_pivot_sequence_is_refined_on_the_device(device_vectors):
    """tests/golden/refinement_case.npz through the C ABI: pp_residual finds the backward error of 0.48, the correction solves
    of pp_refine_begin / _end bring it below 1e-8 -- host containers, device vectors, the deferred form."""
    sc.case_refinement_fixture(make_engine, device_vectors=device_vectors)


def test_adversarial_systems_are_never_returned_inaccurate_on_the_device():
    """>= 400 adversarial systems x 6 factorisations through the C ABI: k_residual / the refinement bracket / the repair path
    of the product (csrc/refine.hip), every handed-out solution against dense algebra."""
    stats = sc.case_adversarial_systems(make_engine, seeds=range(0, 400))
    print('adversarial systems on the device:', stats)


def test_device_vector_kernels_f4():
    """the fused step-statistics kernel, max-norm and step update on device vectors against the restated
    reference formulas (interior_point.py:655-758 fraction_to_the_boundary, :257-266 bound residuals, :619-626)."""
    import torch
    from parapint_amd.algorithms import interior_point as ip
    from parapint_amd.linalg import device_vector_ops as dv
    solver = sc.new_solver(make_engine, 1)
    rng = np.random.default_rng(5)
    for n in (1, 77, 70 * 9200 + 13):
        lb = np.where(rng.random(n) < 0.7, rng.uniform(-2.0, 0.0, n), -np.inf)
        ub = np.where(rng.random(n) < 0.6, rng.uniform(1.0, 3.0, n), np.inf)
        x = rng.uniform(0.1, 0.9, n)
        dx = rng.normal(size=n) * 3.0
        dx[rng.random(n) < 0.05] = 0.0
        zl = np.where(np.isfinite(lb), rng.uniform(0.01, 2.0, n), 0.0)
        zu = np.where(np.isfinite(ub), rng.uniform(0.01, 2.0, n), 0.0)
        dzl, dzu = rng.normal(size=n), rng.normal(size=n)
        tau, mu = 0.995, 0.1
        dev = [torch.from_numpy(a).cuda() for a in (x, dx, lb, ub, zl, dzl, zu, dzu)]
        a_p, a_d, c_l, c_u = dv.step_stats(solver, *dev, tau=tau, barrier=mu)
        ref_p = min(ip._frac_lb(tau, x, dx, lb), ip._frac_ub(tau, x, dx, ub))
        ref_d = min(ip._frac_lb(tau, zl, dzl, np.zeros(n)), ip._frac_lb(tau, zu, dzu, np.zeros(n)))
        fl, fu = np.isfinite(lb), np.isfinite(ub)
        ref_cl = np.abs((x[fl] - lb[fl]) * zl[fl] - mu).max() if fl.any() else 0.0
        ref_cu = np.abs((ub[fu] - x[fu]) * zu[fu] - mu).max() if fu.any() else 0.0
        assert a_p == ref_p and a_d == ref_d          # min / max of identical fp64 expressions: exact
        assert abs(c_l - ref_cl) <= 1e-15 * max(1.0, ref_cl) and abs(c_u - ref_cu) <= 1e-15 * max(1.0, ref_cu)
        assert dv.max_abs(solver, dev[1]) == np.abs(dx).max()
        y = dev[0].clone()
        dv.axpy_(solver, y, a_p, dev[1])
        assert np.abs(y.cpu().numpy() - (x + a_p * dx)).max() <= 4e-16 * (1.0 + np.abs(x).max() + a_p * np.abs(dx).max())  # (one fma)
    # on the solver's own device-resident solution: the step of the synthetic problem is finite and the norms agree
    from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
    from parapint_amd.linalg.comm import SerialComm
    model = SyntheticKKT(70, 30, 2, 10)
    dk = model.build_device_kkt(comm=SerialComm())
    s2 = sc.new_solver(make_engine, 70)
    s2.do_symbolic_factorization(dk)
    dk.set_sources_from_host({ndx: model.block_sources(ndx, 1) for ndx in range(70)})
    s2.do_numeric_factorization(dk)
    xd = s2.do_back_solve(s2.device_vector_from_host(model.build_rhs(comm=SerialComm())))
    t = xd.group_tensors[0]                       # [n][padded batch]; the padded lanes of the solution are zero
    assert float(t[:, 70:].abs().max()) == 0.0
    assert dv.max_abs(s2, t) == float(t[:, :70].abs().max())


@pytest.mark.parametrize('shape', [(2, 3), (6, 4), (70, 5)])
def test_dynamic_time_blocks_dense_schur(shape):
    """f3, small: mapped groups scattered into a dense S (n_c <= 1024); 70 time blocks = a full wave + a ragged one."""
    sc.case_dynamic(make_engine, shape[0], shape[1], expect_block_tridiagonal=False)


def test_dynamic_time_blocks_block_tridiagonal_schur():
    """f3 at BASELINE.json configs[3]'s state dimension: T = 64 time blocks, n_s = 49 (n_c = 6174): ordering by reverse
    Cuthill-McKee, block-tridiagonal S, block LDL^T with Bunch-Kaufman inside the blocks; against the oracle's
    restatement of the reference (sparse S), residual and inertia."""
    solver, model = sc.case_dynamic(make_engine, 64, 49, n_u=2, nfe=4, expect_block_tridiagonal=True)
    gs, G = solver._btd
    assert gs == 2 * 49 and G == 63 and not solver._btd_sequential       # blocks (rho_t, z_t), cyclic reduction
    fast, pivoted = solver._eng.bcr_block_paths()
    assert fast + pivoted == G
    assert fast > 0          # quasi-definite blocks: the unpivoted matrix-core factorisation passes its threshold test


@pytest.mark.parametrize('n_s', [3, 8, 20, 56, 57])
def test_cyclic_reduction_block_sizes(n_s):
    """Block sizes gs = 2 n_s around the tile edges of the unpivoted matrix-core path (6: less than one tile, 16: exactly
    one, 40: ragged, 112: its largest, 114: too large -- Bunch-Kaufman for every block), against the oracle."""
    solver, model = sc.case_dynamic(make_engine, 12, n_s, n_u=2, nfe=2, expect_block_tridiagonal=True, dense_limit=8)
    gs, G = solver._btd
    assert gs == 2 * n_s
    fast, pivoted = solver._eng.bcr_block_paths()
    assert fast + pivoted == G
    if gs > 112:
        assert fast == 0
    else:
        assert fast > 0


@pytest.mark.parametrize('dense_limit', [None, 8])
def test_dynamic_problem_through_the_inertia_correction_loop(dense_limit):
    solver = sc.case_dynamic_regularised(make_engine, dense_limit)
    assert (solver._btd is not None) == (dense_limit is not None)


def test_measurement_switches_select_paths_that_agree():
    """Every round-2 kernel path has a measurement switch that selects its predecessor (one instance per lane, v_readlane
    broadcasts in the dense LDL^T, register-tile Schur update, pattern groups one after the other, scalar block products of
    the cyclic reduction, standalone source gather).  With all of them set the odd-shape soak (residual <= 1e-9, exact
    inertia; one block ... 320 blocks, n_c up to 513) and a block-tridiagonal dynamic case must still pass: the fallbacks
    stay correct."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ('PP_NO_LANE_PAIRS', 'PP_NO_DENSE_DPP', 'PP_NO_SCHUR_MFMA', 'PP_NO_GROUP_STREAMS', 'PP_NO_BCR_MFMA',
              'PP_NO_FUSED_SOURCES', 'PP_BCR_FWD_PHASES', 'PP_NO_EARLY_FORWARD'):
        env[k] = '1'
    env['PYTHONPATH'] = root + os.pathsep + env.get('PYTHONPATH', '')
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'soak_small.py')], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=900)
    text = out.stdout.decode()
    assert out.returncode == 0 and 'small soak ok' in text, text[-3000:]
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import solver_cases as sc\n"
            "from parapint_amd.linalg.hip_schur_complement import HipEngine\n"
            "sc.case_dynamic(lambda: HipEngine(), 64, 49, n_u=2, nfe=4, expect_block_tridiagonal=True)\n"
            "print('dynamic ok')\n") % (root, os.path.join(root, 'tests'))
    out = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert out.returncode == 0 and 'dynamic ok' in out.stdout.decode(), out.stdout.decode()[-3000:]
    # the cyclic reduction with every diagonal block left to Bunch-Kaufman (the path a block takes when the unpivoted
    # factorisation rejects it), matrix-core products for the rest
    env2 = dict(os.environ, PP_NO_BCR_LDL='1', PYTHONPATH=env['PYTHONPATH'])
    code2 = code.replace("print('dynamic ok')", "s = sc.case_dynamic(lambda: HipEngine(), 64, 49, n_u=2, nfe=4, expect_block_tridiagonal=True)[0]\n"
                                               "assert s._eng.bcr_block_paths() == (0, 63)\nprint('dynamic ok')")
    out = subprocess.run([sys.executable, '-c', code2], env=env2, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert out.returncode == 0 and 'dynamic ok' in out.stdout.decode(), out.stdout.decode()[-3000:]
    # ... and a level whose blocks take different paths: with a multiplier bound of 0.7 the unpivoted factorisation
    # rejects the blocks whose largest multiplier exceeds it (0.67 ... 0.82 on this problem), the others keep it
    env3 = dict(os.environ, PP_BCR_LBOUND='0.7', PYTHONPATH=env['PYTHONPATH'])
    code3 = code.replace("print('dynamic ok')", "s = sc.case_dynamic(lambda: HipEngine(), 64, 49, n_u=2, nfe=4, expect_block_tridiagonal=True)[0]\n"
                                               "f, p = s._eng.bcr_block_paths()\nassert f + p == 63 and p > 0, (f, p)\nprint('dynamic ok', f, p)")
    out = subprocess.run([sys.executable, '-c', code3], env=env3, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert out.returncode == 0 and 'dynamic ok' in out.stdout.decode(), out.stdout.decode()[-3000:]


def test_host_boundary_fast_paths_on_the_device():
    """Host blocks in, host vectors out through pp_stage_upload_compact / pp_upload_rhs_rows / pp_download_solution_rows
    (verified index arrays, data rewritten in place, both result-buffer modes), each result against a dense solve."""
    sc.case_boundary_fast_paths(lambda: None)


def test_entries_declared_constant_by_the_producer_on_the_device():
    """declare_constant_entries: pp_stage_upload_verified_begin is given the runs of the variable entries only for rows whose
    staging row holds every entry already; same results as an undeclared solver; check=True reports a declaration that does
    not hold, the periodic full pass heals it."""
    sc.case_constant_entries(lambda: None)


def test_flat_value_vectors_over_the_symbolic_pattern_on_the_device():
    """HostValueMatrix (one flat value vector per block over the pattern object of the symbolic phase): a pattern group is
    staged by one pp_stage_upload_verified_begin call; results equal those of the COO blocks with the same values."""
    sc.case_flat_values(lambda: None)


def test_switching_input_forms_between_factorisations_ends_the_staging_mirror():
    """C-ABI callers may mix input forms: after compact rows were staged (the device mirrors the pinned staging rows and the
    compare-while-staging path sends only pieces that differ), pp_upload_values / a write through pp_raw_buffer overwrite
    the same device buffer.  The next staged factorisation of UNCHANGED host values must send every row again."""
    import ctypes
    import numpy as np
    from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
    from parapint_amd.linalg.comm import SerialComm
    N = 5
    model = SyntheticKKT(N, 3, 8, 2)
    comm = SerialComm()
    solver = sc.new_solver(lambda: None, N)
    kkt = model.build_kkt(comm=comm, iteration=1)
    rhs = model.build_rhs(comm=comm)
    solver.do_symbolic_factorization(kkt)
    solver.do_numeric_factorization(kkt)
    solver.do_numeric_factorization(kkt)            # (second call: verified index arrays, rows mirror the device)
    x_ref = solver.do_back_solve(rhs).flatten().copy()
    eng = solver._eng
    g = solver._groups[0]
    for clobber in ('pp_upload_values', 'pp_raw_buffer'):
        junk = np.full(N * g.nraw, 7.25)
        if clobber == 'pp_upload_values':
            eng.ns.check(eng.lib.pp_upload_values(eng.ns.h, g.gid, junk.ctypes.data, 0), clobber)
        else:
            dev_ptr = eng.lib.pp_raw_buffer(eng.ns.h, g.gid)
            assert dev_ptr
            import torch
            torch.cuda.synchronize()
            hip = ctypes.CDLL('libamdhip64.so')
            assert hip.hipMemcpy(ctypes.c_void_p(dev_ptr), junk.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(junk.nbytes), 1) == 0
        solver.do_numeric_factorization(kkt)        # same host values as the staging rows hold
        x = solver.do_back_solve(rhs).flatten()
        assert np.array_equal(x, x_ref), clobber
        assert sc.scaled_residual(kkt.toarray(), x, rhs.flatten()) <= 1e-10


def test_zero_pivot_test_inside_a_mixed_scale_block_pivot():
    sc.case_mixed_scale_block_pivot(lambda: None)


def test_prefetched_forward_sweep_gives_the_same_solution():
    """solver.prefetch_forward(rhs): the forward sweep enqueued behind the factorisation (beside the dense phase on its own
    stream) and the back-solve that starts at the coupling solve give bit for bit the solution of the plain call order; a
    back-solve with ANOTHER right-hand side falls back to the full sweep; retries re-run the sweep."""
    import torch
    from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
    N = 70
    model = SyntheticKKT(N, 40, 2, 8)
    comm = SerialComm()
    solver = HipSchurComplementLinearSolver({i: None for i in range(N)}, None, comm=comm, result_buffers=0)
    dk = model.build_device_kkt(comm=comm)
    solver.do_symbolic_factorization(dk)
    dk.set_sources_from_host({ndx: model.block_sources(ndx, 1) for ndx in range(N)})
    rhs = solver.device_vector_from_host(model.build_rhs(comm=comm))
    other = solver.device_vector_from_host(model.build_rhs(comm=comm))
    other.group_tensors[0].mul_(2.0)
    solver.do_numeric_factorization(dk)
    x_plain = solver.do_back_solve(rhs)
    x_other = solver.do_back_solve(other)
    solver.prefetch_forward(rhs)
    solver.do_numeric_factorization(dk)
    solver.do_numeric_factorization(dk)                       # (a second factorisation of the iteration: the sweep is redone)
    x_pre = solver.do_back_solve(rhs)
    solver.prefetch_forward(rhs)
    solver.do_numeric_factorization(dk)
    x_fallback = solver.do_back_solve(other)                  # not the announced vector: full sweep
    torch.cuda.synchronize()
    assert torch.equal(x_pre.group_tensors[0], x_plain.group_tensors[0]) and torch.equal(x_pre.coupling, x_plain.coupling)
    assert torch.equal(x_fallback.group_tensors[0], x_other.group_tensors[0]) and torch.equal(x_fallback.coupling, x_other.coupling)
    assert solver._prefetch_rhs is None and solver._forward_done_for is None


def test_prefetched_forward_sweep_of_a_time_staged_problem_gives_the_same_solution():
    """The same for three pattern groups and a block-tridiagonal S (pp_solve_forward_ex: every group's sweep behind the
    factorisation of that group, on its stream, beside the Schur update and the cyclic reduction; the groups of the handle's
    own stream on an auxiliary one): bit for bit the solution of the plain call order, also after the values changed."""
    import torch
    from parapint_amd.examples.performance.schur_complement.dynamic_kkt import SyntheticDynamicKKT
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
    T = 40
    model = SyntheticDynamicKKT(T, 20, 2, 3)
    comm = SerialComm()
    solver = HipSchurComplementLinearSolver({i: None for i in range(T)}, None, comm=comm, result_buffers=0)
    solver._dense_coupling_limit = 8              # (block-tridiagonal S at this small size)
    dk = model.build_device_kkt(comm=comm)
    solver.do_symbolic_factorization(dk)
    assert solver._btd is not None and len(solver._groups) == 3
    rhs = solver.device_vector_from_host(model.build_rhs(comm=comm))
    for it in (1, 2):
        dk.set_sources_from_host({ndx: model.block_sources(ndx, it) for ndx in range(T)})
        solver.do_numeric_factorization(dk)
        x_plain = solver.do_back_solve(rhs)
        solver.prefetch_forward(rhs)
        solver.do_numeric_factorization(dk)
        x_pre = solver.do_back_solve(rhs)
        torch.cuda.synchronize()
        for gid in x_plain.group_tensors:
            assert torch.equal(x_pre.group_tensors[gid], x_plain.group_tensors[gid])
        assert torch.equal(x_pre.coupling, x_plain.coupling)
        assert float(x_plain.coupling.abs().max()) > 0.0
