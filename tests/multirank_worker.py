"""Worker of tests/test_multirank_gloo.py: one rank of a world_size-2 gloo run on CPU.

Rehearses the N>1 host path of HipSchurComplementLinearSolver -- ownership (Q10), the packed
S+status all-reduce, the r_s all-reduce, rank-consistent status -- with the TEST-ONLY host
interpreter standing in for the GPU (the HIP path needs a device; the collectives and the host
logic are identical)."""
import os
import sys

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from hostsim_engine import HostSimEngine  # noqa: E402
from oracle.schur_complement import MPISchurComplementLinearSolver as OracleSC  # noqa: E402
from oracle.subsolvers import ScipyInterface as OracleScipy  # noqa: E402
from parapint_amd.examples.performance.schur_complement.synthetic_kkt import (SyntheticKKT,  # noqa: E402
                                                                              distribute_blocks)
from parapint_amd.linalg.comm import SerialComm, TorchComm  # noqa: E402
from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver  # noqa: E402
from parapint_amd.linalg.results import LinearSolverStatus  # noqa: E402


def main():
    dist.init_process_group('gloo')
    comm = TorchComm()
    rank, size = comm.rank, comm.size
    assert size == 2
    shape = (5, 40, 2, 8)
    N = shape[0]
    local = distribute_blocks(N, rank, size)
    model = SyntheticKKT(*shape, local_blocks=local)
    kkt = model.build_kkt(comm=comm, iteration=2)
    rhs = model.build_rhs(comm=comm)
    solver = HipSchurComplementLinearSolver({i: None for i in local}, None, comm=comm, engine=HostSimEngine())
    assert solver.do_symbolic_factorization(kkt).status == LinearSolverStatus.successful
    assert solver.local_block_indices == local
    assert solver.do_numeric_factorization(kkt).status == LinearSolverStatus.successful
    x = solver.do_back_solve(rhs)
    # single-process oracle on the whole system
    full_model = SyntheticKKT(*shape)
    okkt = full_model.build_kkt(comm=SerialComm(), iteration=2)
    orhs = full_model.build_rhs(comm=SerialComm())
    oracle = OracleSC({i: OracleScipy(compute_inertia=True) for i in range(N)}, OracleScipy(compute_inertia=True))
    oracle.do_symbolic_factorization(okkt)
    oracle.do_numeric_factorization(okkt)
    xo = oracle.do_back_solve(orhs)
    So = oracle.schur_complement.toarray()
    assert np.abs(solver.get_schur_complement() - So).max() <= 1e-9 * np.abs(So).max()
    for ndx in local:
        ref = np.asarray(xo.get_block(ndx))
        assert np.abs(np.asarray(x.get_block(ndx)) - ref).max() <= 1e-8 * np.abs(ref).max()
    for ndx in range(N):
        if ndx not in local:
            assert x.get_block(ndx) is None            # non-local blocks stay unset (mpi_...:390-398)
    assert np.allclose(x.get_block(N), xo.get_block(N), rtol=1e-8, atol=1e-10)
    assert solver.get_inertia() == oracle.get_inertia()
    assert abs(model.check_result(x, comm=comm) - full_model.check_result(xo)) < 1e-8
    # the oracle's own MPI restatement over gloo gives the same S
    o2 = OracleSC({i: OracleScipy(compute_inertia=True) for i in local}, OracleScipy(compute_inertia=True), comm=comm)
    o2.do_symbolic_factorization(kkt)
    o2.do_numeric_factorization(kkt)
    assert np.abs(o2.schur_complement.toarray() - So).max() <= 1e-10 * np.abs(So).max()
    assert o2.get_inertia() == oracle.get_inertia()

    # flat value vectors over the pattern object (HostValueMatrix): row i = the i-th block THIS rank owns; with the
    # producer's declaration of its constant entries
    from parapint_amd.sparse.host_value_matrix import HostValueMatrix
    solver.declare_constant_entries(model.constant_entries())
    for it in (4, 5):
        assert solver.do_numeric_factorization(HostValueMatrix(kkt, model.flat_values(iteration=it))).status == \
            LinearSolverStatus.successful
        xf = solver.do_back_solve(rhs)
        k_it = model.build_kkt(comm=comm, iteration=it)
        assert solver.do_numeric_factorization(k_it).status == LinearSolverStatus.successful
        xk = solver.do_back_solve(rhs)
        for ndx in local:
            assert np.array_equal(np.asarray(xf.get_block(ndx)), np.asarray(xk.get_block(ndx)))
        assert np.array_equal(np.asarray(xf.get_block(N)), np.asarray(xk.get_block(N)))
    # a declaration that does not hold on ONE rank (check=True): that rank's staging fails with a status BEFORE the
    # collective that agrees on a changed pattern -- it must still join it, and both ranks must report the error
    # (the engine with the host-boundary entry points: the check sits on the verified fast path)
    from hostsim_engine import HostSimBoundaryEngine
    sv = HipSchurComplementLinearSolver({i: None for i in local}, None, comm=comm, engine=HostSimBoundaryEngine())
    assert sv.do_symbolic_factorization(kkt).status == LinearSolverStatus.successful
    sv.declare_constant_entries(model.constant_entries(), check=True)
    for it in (5, 6):
        assert sv.do_numeric_factorization(model.build_kkt(comm=comm, iteration=it)).status == LinearSolverStatus.successful
    k_bad = model.build_kkt(comm=comm, iteration=7)
    if rank == 1:
        Kb = k_bad.get_block(local[0], local[0])
        Kb.data = Kb.data.copy()
        e = int(np.flatnonzero(model._row > model._col)[3])
        m = int(np.flatnonzero((model._row == model._col[e]) & (model._col == model._row[e]))[0])
        Kb.data[[e, m]] *= 1.25
    res = sv.do_numeric_factorization(k_bad, raise_on_error=False)
    assert res.status == LinearSolverStatus.error, (rank, res.status)
    assert sv.do_numeric_factorization(k_bad).status == LinearSolverStatus.successful     # (the values were staged in full)
    xv = sv.do_back_solve(rhs)
    solver.declare_constant_entries(None)
    assert solver.do_numeric_factorization(k_bad).status == LinearSolverStatus.successful
    xw = solver.do_back_solve(rhs)
    assert np.array_equal(np.asarray(xv.get_block(N)), np.asarray(xw.get_block(N)))
    solver.declare_constant_entries(None)
    assert solver.do_numeric_factorization(kkt).status == LinearSolverStatus.successful

    # status agreement: a singular block on rank 1 only must be reported by both ranks
    if rank == 1:
        from scipy.sparse import coo_matrix
        K = kkt.get_block(local[0], local[0]).tocoo()
        d = K.data.copy()
        d[:] = 0.0
        kkt.set_block(local[0], local[0], coo_matrix((d, (K.row, K.col)), shape=K.shape))
    res = solver.do_numeric_factorization(kkt, raise_on_error=False)
    assert res.status == LinearSolverStatus.singular
    try:
        solver.do_numeric_factorization(kkt, raise_on_error=True)
        raise AssertionError('expected RuntimeError')
    except RuntimeError:
        pass
    # a host-side failure on ONE rank (device storage over budget: status 1) must come back as a status on BOTH ranks --
    # the failing rank still joins the all-reduce with a zero contribution whose tail carries the failure -- and the
    # caller's reallocation loop (interior_point.py:634-652), run identically on both ranks, then succeeds
    kkt2 = model.build_kkt(comm=comm, iteration=1)
    eng = HostSimEngine()
    s2 = HipSchurComplementLinearSolver({i: None for i in local}, None, comm=comm, engine=eng)
    s2.do_symbolic_factorization(kkt2)
    if rank == 1:
        eng.set_memory_budget(eng.required_bytes() // 3)
    for count in range(5):
        res = s2.do_numeric_factorization(matrix=kkt2, raise_on_error=False)
        if res.status == LinearSolverStatus.not_enough_memory:
            s2.increase_memory_allocation(2)
        else:
            break
    assert res.status == LinearSolverStatus.successful and count == 2, (rank, res.status, count)
    x2 = s2.do_back_solve(rhs)
    okkt2 = full_model.build_kkt(comm=SerialComm(), iteration=1)
    oracle.do_numeric_factorization(okkt2)
    xo2 = oracle.do_back_solve(full_model.build_rhs(comm=SerialComm()))
    assert np.allclose(x2.get_block(N), xo2.get_block(N), rtol=1e-8, atol=1e-10)
    # dynamic (time-staged) problem over two ranks: mapped groups, the clique table all-reduce, block-tridiagonal S
    from parapint_amd.examples.performance.schur_complement.dynamic_kkt import SyntheticDynamicKKT
    T = 7
    dyn = SyntheticDynamicKKT(T, 3, 2, 2, local_blocks=distribute_blocks(T, rank, size))
    dk = dyn.build_kkt(comm=comm, iteration=1)
    drhs = dyn.build_rhs(comm=comm)
    s3 = HipSchurComplementLinearSolver({i: None for i in dyn.local_blocks}, None, comm=comm, engine=HostSimEngine())
    s3._dense_coupling_limit = 4
    assert s3.do_symbolic_factorization(dk).status == LinearSolverStatus.successful
    assert s3._btd is not None
    assert s3.do_numeric_factorization(dk).status == LinearSolverStatus.successful
    x3 = s3.do_back_solve(drhs)
    full_dyn = SyntheticDynamicKKT(T, 3, 2, 2)
    Kd = full_dyn.build_kkt(comm=SerialComm(), iteration=1).tocoo().toarray()
    xd = np.linalg.solve(Kd, full_dyn.build_rhs(comm=SerialComm()).flatten())
    off = np.concatenate([[0], np.cumsum([full_dyn.block_dim(t) for t in range(T)])])
    for t in dyn.local_blocks:
        assert np.abs(x3.get_block(t).flatten() - xd[off[t]:off[t + 1]]).max() <= 1e-8 * np.abs(xd).max()
    assert np.abs(np.asarray(x3.get_block(T)) - xd[off[T]:]).max() <= 1e-8 * np.abs(xd).max()
    ev = np.linalg.eigvalsh(Kd)
    assert s3.get_inertia() == (int((ev > 0).sum()), int((ev < 0).sum()), 0)
    # the reference's own MPI test (test_mpi_explicit_schur_complement.py:22-115): the 8x8 system with Q = [[0, 0], [0, 1]],
    # ownership (ndx - rank) % size == 0, solution and inertia against the golden vectors of the reference's solver, a
    # second numeric factorisation + solve on the same object; symmetric variant through the plain route, the unsymmetric
    # original through ScipyInterface objects
    from scipy.sparse import coo_matrix as _coo8
    from parapint_amd.sparse.block_containers import MPIBlockMatrix, MPIBlockVector
    from parapint_amd.linalg import ScipyInterface as _HipScipy
    golden = np.load(os.path.join(HERE, 'golden', 'reference_vectors.npz'))
    own8 = np.array([[-1] * 4 for _ in range(4)])
    owners = [0, 1, 0]                     # (ndx - rank) % 2 == 0
    for i in range(3):
        own8[i, i] = owners[i]
        own8[3, i] = owners[i]
    mine8 = [i for i in range(3) if owners[i] == rank]
    for variant in ('sym', 'unsym'):
        if variant == 'sym':
            ks = [np.array([[1, 0.5], [0.5, 1]]), np.eye(2), np.array([[1, 1], [1, 3.]])]
        else:
            ks = [np.array([[1, 1], [0, 1.]]), np.eye(2), np.array([[1, 0], [1, 1.]])]
        a8 = [np.array([[0, -1], [0, 0.]]), np.array([[-1, 0], [0, -1.]]), np.array([[0, 0], [-1, 0.]])]
        A8 = MPIBlockMatrix(4, 4, own8, comm)
        r8 = MPIBlockVector(4, np.array(owners + [-1]), comm)
        for i in mine8:
            A8.set_block(i, i, _coo8(ks[i]))
            A8.set_block(3, i, _coo8(a8[i]))
        A8.set_block(3, 3, _coo8(np.array([[0, 0], [0, 1.0]])))
        for i in range(4):
            A8.set_row_size(i, 2)
            A8.set_col_size(i, 2)
        vals8 = ([1, 0], [0, 0], [0, 1], [1, 1])
        for i in mine8 + [3]:
            r8.set_block(i, np.array(vals8[i], dtype=np.double))
        e8 = HostSimEngine()
        subs = {i: (_HipScipy(compute_inertia=True, engine=e8) if variant == 'unsym' else None) for i in mine8}
        s8 = HipSchurComplementLinearSolver(subs, _HipScipy(compute_inertia=True, engine=e8) if variant == 'unsym' else None,
                                            comm=comm, engine=e8)
        assert s8.do_symbolic_factorization(A8).status == LinearSolverStatus.successful
        assert s8.local_block_indices == mine8
        key = 'b8_%s_mpi' % variant
        for _ in range(2):
            assert s8.do_numeric_factorization(A8).status == LinearSolverStatus.successful
            x8 = s8.do_back_solve(r8)
            for i in mine8 + [3]:
                assert np.allclose(np.asarray(x8.get_block(i)), golden[key + '_x'][2 * i:2 * i + 2], rtol=1e-10, atol=1e-10), (variant, i)
        assert np.allclose(s8.get_schur_complement(), golden[key + '_S'], rtol=1e-12, atol=1e-12)
        if variant == 'sym':
            assert s8.get_inertia() == tuple(golden[key + '_inertia'])
    # general-LU semantics (general_blocks.py): ScipyInterface objects as sub-solvers, ONE diagonal block -- on rank 1 -- not
    # symmetric: both ranks must take the embedding (the decision is an all-reduce), and the solution is that of the oracle's
    # LU sub-solvers on the same matrix
    from parapint_amd.linalg import ScipyInterface as HipScipy
    kg = model.build_kkt(comm=comm, iteration=3)
    okg = full_model.build_kkt(comm=SerialComm(), iteration=3)
    odd = distribute_blocks(N, 1, size)[0]
    up = model._row < model._col
    for mat, mine in ((kg, rank == 1), (okg, True)):
        if mine:
            Kg = mat.get_block(odd, odd)
            Kg.data = Kg.data.copy()
            Kg.data[up] *= 1.5
    e5 = HostSimEngine()
    s5 = HipSchurComplementLinearSolver({i: HipScipy(engine=e5) for i in local}, HipScipy(engine=e5), comm=comm, engine=e5)
    assert s5.do_symbolic_factorization(kg).status == LinearSolverStatus.successful
    assert s5._general_mode is True
    assert s5.do_numeric_factorization(kg).status == LinearSolverStatus.successful
    x5 = s5.do_back_solve(rhs)
    oracle.do_numeric_factorization(okg)
    xo5 = oracle.do_back_solve(orhs)
    for ndx in local:
        ref = np.asarray(xo5.get_block(ndx))
        assert np.abs(np.asarray(x5.get_block(ndx)) - ref).max() <= 1e-8 * np.abs(ref).max()
    assert np.allclose(x5.get_block(N), xo5.get_block(N), rtol=1e-8, atol=1e-10)
    assert type(x5) is type(rhs) and all(x5.get_block(ndx) is None for ndx in range(N) if ndx not in local)
    # ... and the symmetric matrix again through the same object: the symmetric path, with its inertia
    assert s5.do_numeric_factorization(model.build_kkt(comm=comm, iteration=2)).status == LinearSolverStatus.successful
    assert s5._general_mode is False
    oracle.do_numeric_factorization(okkt)
    assert s5.get_inertia() == oracle.get_inertia()
    assert np.allclose(s5.do_back_solve(rhs).get_block(N), xo.get_block(N), rtol=1e-8, atol=1e-10)
    dist.barrier()
    dist.destroy_process_group()
    print('rank %d ok' % rank)


if __name__ == '__main__':
    main()
