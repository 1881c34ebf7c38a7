"""The device interior-point loop over a few problem sizes and seeds (robustness: convergence, pivot-order refreshes,
regularisation retries, iteration counts): python tools/ip_soak.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from parapint_amd.algorithms.device_interior_point import ip_solve_device
    from parapint_amd.algorithms.interior_point import IPOptions, InteriorPointStatus
    from parapint_amd.examples.stochastic_qp import c3_stochastic_qp, random_stochastic_qp
    from parapint_amd.interfaces.schur_complement.device_sc_ip_interface import DeviceStochasticQPInterface
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
    cases = [('c3', 128, 1000, 200, 1), ('c3', 1024, 1000, 200, 2), ('c3', 1024, 1000, 200, 3), ('c3', 300, 400, 100, 4),
             ('c3', 64, 2000, 400, 5), ('rand', 1024, 120, 10, 7), ('rand', 333, 60, 6, 8)]
    for kind, N, a, b, seed in cases:
        if kind == 'c3':
            qps, fs = c3_stochastic_qp(N, n_q=a, m=4, n_theta=b, seed=seed)
        else:
            qps, fs = random_stochastic_qp(N, n=a, n_fs=b, n_eq=a // 4, n_ineq=a // 3, seed=seed)
        it = DeviceStochasticQPInterface(qps, fs)
        opt = IPOptions()
        solver = HipSchurComplementLinearSolver({i: None for i in range(N)}, None, comm=SerialComm(), result_buffers=2)
        opt.linalg.solver = solver
        hist, stats = [], {}
        t0 = time.time()
        try:
            status, iters = ip_solve_device(it, opt, history=hist, stats=stats)
            ok = status == InteriorPointStatus.optimal
            err = None
        except Exception as e:       # noqa: BLE001 (soak: report and go on)
            ok, iters, err = False, len(hist), repr(e)[:200]
        torch.cuda.synchronize()
        print(json.dumps({'case': [kind, N, a, b, seed], 'converged': ok, 'iterations': iters, 'error': err,
                          'ms_per_iteration': 1e3 * stats.get('loop_s', 0.0) / max(iters, 1), 'wall_s': time.time() - t0,
                          'refreshes': solver.pivot_order_refreshes, 'retries': solver.diagonal_shift_refactorizations,
                          'final': list(hist[-1][:3]) if hist else None}), flush=True)
        del it, solver, opt, qps


if __name__ == '__main__':
    main()
