"""Worker of tests/test_device_ip.py::test_two_rank_device_loop: one rank of a world_size-2 gloo run of the
interior-point loop with rank-distributed device-resident iterates (numpy engines standing in for the GPU: the collectives
-- S, r_s, the two all-gathers of the step -- and the host logic are those of the device run)."""
import os
import sys

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

GPU = '--gpu' in sys.argv      # the real kernels, both ranks on the visible device(s), collectives through gloo (rehearsal)
if not GPU:
    from hostsim_engine import HostSimDeviceEngine  # noqa: E402
from parapint_amd.algorithms.device_interior_point import ip_solve_device  # noqa: E402
from parapint_amd.algorithms.interior_point import IPOptions, InteriorPointStatus  # noqa: E402
from parapint_amd.examples.stochastic_qp import random_stochastic_qp  # noqa: E402
from parapint_amd.interfaces.schur_complement.device_sc_ip_interface import DeviceStochasticQPInterface  # noqa: E402
from parapint_amd.linalg.comm import SerialComm, TorchComm  # noqa: E402
from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver  # noqa: E402


def mixed_scenarios():
    """Seven scenarios of two sparsity patterns (different sizes), interleaved: both ranks hold both patterns."""
    a, fa = random_stochastic_qp(4, n=24, n_fs=4, n_eq=6, n_ineq=8, seed=2)
    b, fb = random_stochastic_qp(3, n=17, n_fs=4, n_eq=5, n_ineq=3, seed=5)
    qps = [a[0], a[1], b[0], b[1], a[2], a[3], b[2]]          # (dealt round-robin over two ranks: a b a b / a b a)
    fs = [fa[0], fa[1], fb[0], fb[1], fa[2], fa[3], fb[2]]
    return qps, fs


def run(comm, qps, fs):
    it = DeviceStochasticQPInterface(qps, fs, comm=comm)
    opt = IPOptions()
    opt.linalg.solver = HipSchurComplementLinearSolver({i: None for i in it.local}, None, comm=comm,
                                                       engine=None if GPU else HostSimDeviceEngine(), result_buffers=2 if GPU else 0)
    hist = []
    status, iters = ip_solve_device(it, opt, history=hist)
    assert status == InteriorPointStatus.optimal
    return it, hist


def main():
    dist.init_process_group('gloo')
    if GPU:
        import torch
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count())
    comm = TorchComm()
    assert comm.size == 2
    qps, fs = mixed_scenarios()
    it, hist = run(comm, qps, fs)
    assert len(it.pattern_groups) == 2 and len(it.states) == 2
    ref, ref_hist = run(SerialComm(), qps, fs)          # the same loop on one rank
    assert len(hist) == len(ref_hist)
    for a, b in zip(hist, ref_hist):
        assert np.allclose(a[:6], b[:6], rtol=1e-6, atol=1e-9), (a, b)
    assert np.abs(it.first_stage_solution() - ref.first_stage_solution()).max() <= 1e-8
    for ndx in it.local:
        assert np.abs(it.scenario_primals(ndx) - ref.scenario_primals(ndx)).max() <= 1e-7
    # every rank took the same decisions from the same numbers
    mine = np.array([v for row in hist for v in row[:6]])
    both = comm.allgather(mine)
    assert np.array_equal(both[0], both[1])
    print('rank %d ok' % comm.rank)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
