"""The device interior-point loops of time-staged problems over sizes around the limits of the cyclic reduction's unpivoted
path (block size 2 n_s <= 112), few / many time blocks, quadratic (examples/dynamics_qp.py) and nonlinear (examples/burgers.py)
models: convergence, iteration counts, which path the diagonal blocks of S took, and for the small cases the objective of the
host producer over the same solver class.  python tools/dynamic_soak.py   (run on the GPU box)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from parapint_amd.algorithms.device_interior_point import ip_solve_device
    from parapint_amd.algorithms.interior_point import IPOptions, InteriorPointStatus, ip_solve
    from parapint_amd.examples import burgers, dynamics_qp as dq
    from parapint_amd.interfaces.schur_complement.device_sc_ip_interface import DeviceDynamicQPInterface
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver

    def solver(T):
        return HipSchurComplementLinearSolver({t: None for t in range(T)}, None, comm=SerialComm(), result_buffers=2)

    ok = True
    qp_cases = [(8, 5, 1, 3), (33, 12, 2, 4), (64, 30, 3, 4), (128, 49, 2, 8), (200, 56, 2, 4), (40, 57, 2, 3), (3, 49, 2, 6),
                (257, 20, 3, 2)]
    for T, ns, nu, nfe in qp_cases:
        args = dict(nfe_per_block=nfe, n_states=ns, n_controls=nu, nu=0.15 / (ns + 1) ** 2 * T * nfe)
        blocks = dq.DiffusionControl.time_blocks(0.0, 1.0, T, **args)
        it = DeviceDynamicQPInterface(blocks)
        opt = IPOptions()
        opt.linalg.solver = solver(T)
        stats = {}
        t0 = time.time()
        status, iters = ip_solve_device(it, opt, stats=stats)
        torch.cuda.synchronize()
        row = dict(problem='diffusion QP', time_blocks=T, states=ns, controls=nu, steps=nfe, status=str(status), iterations=iters,
                   ms_per_iteration=1e3 * stats['loop_s'] / max(iters, 1), wall_s=time.time() - t0,
                   objective=it.evaluate_objective(), btd=opt.linalg.solver._btd,
                   bcr_paths=list(opt.linalg.solver._eng.bcr_block_paths()) if opt.linalg.solver._btd else None,
                   refreshes=opt.linalg.solver.pivot_order_refreshes)
        good = status == InteriorPointStatus.optimal
        if T * ns <= 2000:          # host producer (the restated reference interface) over the same solver class
            host = dq.main(solver(T), 0.0, 1.0, T, **args)
            row['host_objective'] = host.evaluate_objective()
            good = good and abs(row['host_objective'] - row['objective']) <= 1e-7 * max(1.0, abs(row['objective']))
        row['ok'] = bool(good)
        ok = ok and good
        print(json.dumps(row), flush=True)
    for T, nfe_x, per in [(8, 20, 4), (32, 50, 8), (64, 30, 16), (5, 57, 3)]:
        it = burgers.device_interface(nfe_x, T * per, T)
        opt = IPOptions()
        opt.linalg.solver = solver(T)
        stats = {}
        t0 = time.time()
        status, iters = ip_solve_device(it, opt, stats=stats)
        torch.cuda.synchronize()
        row = dict(problem='Burgers', time_blocks=T, nfe_x=nfe_x, steps=per, status=str(status), iterations=iters,
                   ms_per_iteration=1e3 * stats['loop_s'] / max(iters, 1), wall_s=time.time() - t0,
                   objective=it.evaluate_objective(), btd=opt.linalg.solver._btd,
                   bcr_paths=list(opt.linalg.solver._eng.bcr_block_paths()) if opt.linalg.solver._btd else None)
        good = status == InteriorPointStatus.optimal
        if T * nfe_x <= 400:
            host = burgers.main(solver(T), nfe_x=nfe_x, nfe_t=T * per, nblocks=T)
            row['host_objective'] = host.evaluate_objective()
            good = good and abs(row['host_objective'] - row['objective']) <= 1e-7 * max(1.0, abs(row['objective']))
        row['ok'] = bool(good)
        ok = ok and good
        print(json.dumps(row), flush=True)
    print('dynamic soak ok' if ok else 'dynamic soak FAILED')
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
