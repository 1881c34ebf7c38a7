"""The interior-point loop with device-resident iterates on the C3-shaped stochastic QP (1024 scenarios x 5000 primal
variables x 200 first-stage variables by default): iterations, wall time of set-up and of the loop, the solver's phase
times inside the loop, torch operators dispatched per iteration.  python tools/ip_c3.py [scenarios] [n_q] [n_theta]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    n_q = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    n_t = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    import torch
    from parapint_amd.algorithms.device_interior_point import ip_solve_device
    from parapint_amd.algorithms.interior_point import IPOptions, InteriorPointStatus
    from parapint_amd.examples.stochastic_qp import c3_stochastic_qp
    from parapint_amd.interfaces.schur_complement.device_sc_ip_interface import DeviceStochasticQPInterface
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
    t0 = time.time()
    qps, fs = c3_stochastic_qp(N, n_q=n_q, m=4, n_theta=n_t, seed=1)
    t_gen = time.time() - t0
    out = {}
    for rep in range(2):
        it = DeviceStochasticQPInterface(qps, fs)
        opt = IPOptions()
        solver = HipSchurComplementLinearSolver({i: None for i in range(N)}, None, comm=SerialComm(), result_buffers=2)
        opt.linalg.solver = solver
        hist, stats = [], {}
        lib, h = solver._eng.lib, solver._eng.ns.h
        torch.cuda.synchronize()
        t0 = time.time()
        status, iters = ip_solve_device(it, opt, history=hist, stats=stats)
        torch.cuda.synchronize()
        wall = time.time() - t0
        out = {'scenarios': N, 'block_dim': it.pattern_groups[0].nb, 'n_coupling': it.nfs, 'iterations': iters,
               'converged': status == InteriorPointStatus.optimal, 'generate_s': t_gen, 'wall_s': wall,
               'setup_s': stats['setup_s'], 'loop_s': stats['loop_s'], 'ms_per_iteration': 1e3 * stats['loop_s'] / max(iters, 1),
               'it_per_s_loop': iters / stats['loop_s'], 'it_per_s_whole_call': iters / wall,
               'torch_ops_per_iteration': (stats['torch_ops'] or 0) / max(iters, 1),
               'iteration_ms': [round(1e3 * v, 3) for v in stats['iteration_s']],
               'refreshes': solver.pivot_order_refreshes, 'retries': solver.diagonal_shift_refactorizations,
               'final': list(hist[-1][:3]), 'plan': {k: solver.plan_stats[0][k] for k in ('n', 'n_levels', 'u_doubles')}}
        print(json.dumps(out), flush=True)
    # where an iteration goes: the solver's phases (HIP events) over a second run of the same loop
    import ctypes
    import numpy as np
    it = DeviceStochasticQPInterface(qps, fs)
    opt = IPOptions()
    solver = HipSchurComplementLinearSolver({i: None for i in range(N)}, None, comm=SerialComm(), result_buffers=2)
    opt.linalg.solver = solver
    lib, h = solver._eng.lib, solver._eng.ns.h
    dk = it.device_kkt_matrix()
    solver.do_symbolic_factorization(dk)
    lib.pp_profile(h, 1)
    # (re-run through the public entry: the symbolic phase is repeated, the profile covers all of it)
    stats = {}
    status, iters = ip_solve_device(it, opt, stats=stats)
    ms = np.zeros(16); launches = np.zeros(16, dtype=np.int32); calls = np.zeros(16, dtype=np.int32)
    lib.pp_phase_times(h, ms.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), launches.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                       calls.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    lib.pp_profile(h, 0)
    print(json.dumps({'solver_kernel_ms_per_iteration': float(ms.sum()) / max(iters, 1), 'phase_ms_total': ms.tolist(),
                      'phase_calls': calls.tolist(), 'iterations': iters,
                      'loop_ms_per_iteration_with_profiling': 1e3 * stats['loop_s'] / max(iters, 1)}), flush=True)


if __name__ == '__main__':
    main()
