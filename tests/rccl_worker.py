"""Worker of tests/test_hip_solver.py::test_rccl_collectives_on_solver_buffers: ONE rank, backend nccl (= RCCL).

A one-GPU box cannot host two RCCL ranks, so this run issues the two data-path all-reduces (S + status tail,
r_s) over a one-rank RCCL group on the solver's own device buffers and stream: a sum over one rank is the
identity, so the solve must still match the oracle -- which checks the buffer binding, dtype/size and the
stream ordering between the solver's kernels and RCCL that the N > 1 runs rely on."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from oracle.schur_complement import SchurComplementLinearSolver as OracleSC  # noqa: E402
from oracle.subsolvers import ScipyInterface as OracleScipy  # noqa: E402
from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT  # noqa: E402
from parapint_amd.linalg.comm import SerialComm, TorchComm  # noqa: E402
from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver  # noqa: E402
from parapint_amd.linalg.results import LinearSolverStatus  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    comm = TorchComm()
    assert comm.size == 1 and comm.device_collectives
    comm.always_reduce = True
    calls = []
    plain = comm.allreduce_sum_tensor_

    def counted(t):
        calls.append((t.numel(), t.is_cuda))
        return plain(t)
    comm.allreduce_sum_tensor_ = counted
    shape = (70, 40, 2, 8)          # two 64-instance chunks, the second one ragged
    N = shape[0]
    model = SyntheticKKT(*shape)
    solver = HipSchurComplementLinearSolver({i: None for i in range(N)}, None, comm=comm)
    for it in (0, 3):
        kkt = model.build_kkt(comm=comm, iteration=it)
        rhs = model.build_rhs(comm=comm)
        if it == 0:
            assert solver.do_symbolic_factorization(kkt).status == LinearSolverStatus.successful
        assert solver.do_numeric_factorization(kkt).status == LinearSolverStatus.successful
        x = solver.do_back_solve(rhs)
        okkt = model.build_kkt(comm=SerialComm(), iteration=it)
        oracle = OracleSC({i: OracleScipy(compute_inertia=True) for i in range(N)}, OracleScipy(compute_inertia=True))
        oracle.do_symbolic_factorization(okkt)
        oracle.do_numeric_factorization(okkt)
        xo = oracle.do_back_solve(model.build_rhs(comm=SerialComm()))
        for ndx in range(N + 1):
            ref = np.asarray(xo.get_block(ndx))
            assert np.abs(np.asarray(x.get_block(ndx)) - ref).max() <= 1e-8 * max(1.0, np.abs(ref).max())
        assert solver.get_inertia() == oracle.get_inertia()
    nc = shape[3]
    assert calls.count((nc * nc + 8, True)) == 2 and calls.count((nc, True)) == 2, calls
    # the same exchanges enqueued by the library itself (include/parapint_hip.h: pp_comm_init, pp_allreduce_schur,
    # pp_allreduce_rs): communicator from a unique id, RCCL on the handle's stream, no torch.distributed call in between
    comm.direct_rccl = True
    del calls[:]
    solver2 = HipSchurComplementLinearSolver({i: None for i in range(N)}, None, comm=comm)
    kkt = model.build_kkt(comm=comm, iteration=3)
    rhs = model.build_rhs(comm=comm)
    assert solver2.do_symbolic_factorization(kkt).status == LinearSolverStatus.successful
    assert solver2.do_numeric_factorization(kkt).status == LinearSolverStatus.successful
    x2 = solver2.do_back_solve(rhs)
    assert solver2._eng.lib.pp_comm_size(solver2._eng.ns.h) == 1
    assert not calls, calls                               # torch.distributed was not asked to reduce anything
    # ... the a-posteriori check of the back-solve included: the sums of its coupling rows and the per-rank block results
    # went through ONE all-reduce of the library's communicator (pp_residual with coupling_on_device = 2)
    assert solver2.last_residual is not None and solver2.last_residual <= 1e-12 and solver2._eng.lib.pp_comm_size(solver2._eng.ns.h) == 1
    assert solver.last_residual is not None and abs(solver2.last_residual - solver.last_residual) <= 1e-13
    for ndx in range(N + 1):
        assert np.array_equal(np.asarray(x2.get_block(ndx)), np.asarray(x.get_block(ndx)))
    assert solver2.get_inertia() == solver.get_inertia()
    # librccl that cannot be opened on some rank (here: the id call made to fail) is agreed BEFORE the collective set-up of
    # the communicator and sends the handle to the torch.distributed collectives -- nobody is left inside ncclCommInitRank
    solver3 = HipSchurComplementLinearSolver({i: None for i in range(N)}, None, comm=comm)
    lib3 = solver3._eng.lib
    real_id = lib3.pp_comm_unique_id
    lib3.pp_comm_unique_id = lambda *a: 3
    try:
        two = type('TwoRanks', (), {})()          # what a two-rank group asks of the engine (its collectives stubbed: one process)
        two.rank, two.size, two.device_collectives, two.direct_rccl, two._group = 0, 2, True, None, None
        two._dist = type('D', (), {'all_reduce': staticmethod(lambda *a, **k: None),
                                   'broadcast': staticmethod(lambda *a, **k: None), 'ReduceOp': dist.ReduceOp})()
        assert solver3._eng._direct_rccl(two) is False and solver3._eng._rccl_unavailable is True
    finally:
        lib3.pp_comm_unique_id = real_id
    comm.direct_rccl = None
    assert solver3.do_symbolic_factorization(kkt).status == LinearSolverStatus.successful
    assert solver3.do_numeric_factorization(kkt).status == LinearSolverStatus.successful
    x3 = solver3.do_back_solve(rhs)
    for ndx in range(N + 1):
        assert np.array_equal(np.asarray(x3.get_block(ndx)), np.asarray(x.get_block(ndx)))
    comm.direct_rccl = True
    # the interior-point loop with its two all-gathers issued by the library (pp_comm_allgather) on the one-rank group:
    # same iterates as the loop without collectives
    from parapint_amd.algorithms.device_interior_point import ip_solve_device
    from parapint_amd.algorithms.interior_point import IPOptions, InteriorPointStatus
    from parapint_amd.examples.stochastic_qp import random_stochastic_qp
    from parapint_amd.interfaces.schur_complement.device_sc_ip_interface import DeviceStochasticQPInterface
    qps, fs = random_stochastic_qp(70, seed=3)
    runs = []
    for c in (comm, SerialComm()):
        it = DeviceStochasticQPInterface(qps, fs, comm=c)
        opt = IPOptions()
        opt.linalg.solver = HipSchurComplementLinearSolver({i: None for i in range(70)}, None, comm=c, result_buffers=2)
        hist = []
        status, iters = ip_solve_device(it, opt, history=hist)
        assert status == InteriorPointStatus.optimal
        runs.append((iters, it.first_stage_solution(), opt.linalg.solver))
    assert runs[0][2]._eng.lib.pp_comm_size(runs[0][2]._eng.ns.h) == 1
    assert runs[0][0] == runs[1][0] and np.array_equal(runs[0][1], runs[1][1])
    dist.barrier()
    dist.destroy_process_group()
    print('rccl one-rank ok')


if __name__ == '__main__':
    main()
