"""Where the set-up of the device interior-point loop goes (symbolic phase, value maps, scenario data to the device):
cProfile of everything before the first iteration at C3 dimensions.  python tools/ip_setup_profile.py [scenarios]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    import torch
    from parapint_amd.algorithms.interior_point import IPOptions, try_factorization_and_reallocation
    from parapint_amd.examples.stochastic_qp import c3_stochastic_qp
    from parapint_amd.interfaces.schur_complement.device_sc_ip_interface import DeviceStochasticQPInterface
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
    qps, fs = c3_stochastic_qp(N, seed=1)
    for rep in range(2):
        pr = cProfile.Profile()
        torch.cuda.synchronize()
        t0 = time.time()
        pr.enable()
        it = DeviceStochasticQPInterface(qps, fs)
        t1 = time.time()
        solver = HipSchurComplementLinearSolver({i: None for i in range(N)}, None, comm=SerialComm(), result_buffers=2)
        dk = it.device_kkt_matrix()
        t2 = time.time()
        try_factorization_and_reallocation(dk, solver, 2, 5, 'symbolic', None)
        t3 = time.time()
        it.attach(solver, dk)
        it.set_barrier_parameter(0.1)
        it.take_step()
        it.check_convergence(1.0)
        torch.cuda.synchronize()
        t4 = time.time()
        pr.disable()
        print('rep %d: interface %.3f  kkt matrix %.3f  symbolic %.3f  attach + first measures %.3f  total %.3f s' %
              (rep, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0), flush=True)
    pstats.Stats(pr).sort_stats('cumulative').print_stats(22)


if __name__ == '__main__':
    main()
