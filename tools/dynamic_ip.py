"""The interior-point loop of a time-staged problem (parapint_amd/examples/dynamics_qp.py) with device-resident iterates
(DeviceDynamicQPInterface) and, for comparison, with the host producer over the same HIP solver class.
python tools/dynamic_ip.py [time blocks] [states] [controls] [steps per block]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    nu = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    nfe = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    import torch
    from parapint_amd.algorithms.device_interior_point import ip_solve_device
    from parapint_amd.algorithms.interior_point import IPOptions
    from parapint_amd.examples import dynamics_qp as dq
    from parapint_amd.interfaces.schur_complement.device_sc_ip_interface import DeviceDynamicQPInterface
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
    args = dict(nfe_per_block=nfe, n_states=ns, n_controls=nu, nu=0.15 / (ns + 1) ** 2 * T * nfe)    # (stable explicit Euler)
    blocks = dq.DiffusionControl.time_blocks(0.0, 1.0, T, **args)
    out = dict(time_blocks=T, states=ns, controls=nu, steps_per_block=nfe, block_dim=int(blocks[1][0].n), n_coupling=2 * ns * (T - 1))
    for rep in range(2):
        it = DeviceDynamicQPInterface(blocks)
        opt = IPOptions()
        opt.linalg.solver = HipSchurComplementLinearSolver({t: None for t in range(T)}, None, comm=SerialComm(), result_buffers=2)
        stats = {}
        torch.cuda.synchronize()
        t0 = time.time()
        status, iters = ip_solve_device(it, opt, stats=stats)
        torch.cuda.synchronize()
        out['device_producer'] = dict(status=str(status), iterations=iters, loop_seconds=stats['loop_s'], setup_seconds=stats['setup_s'],
                                      ms_per_iteration=1e3 * stats['loop_s'] / max(iters, 1), wall_seconds=time.time() - t0,
                                      torch_ops=stats['torch_ops'], torch_op_names=stats['torch_op_names'],
                                      objective=it.evaluate_objective())
    # where an iteration goes: one more run with the library's phase events on (no overlap between phases while they are)
    import ctypes
    import numpy as np
    it = DeviceDynamicQPInterface(blocks)
    opt = IPOptions()
    opt.linalg.solver = HipSchurComplementLinearSolver({t: None for t in range(T)}, None, comm=SerialComm(), result_buffers=2)
    lib, h = opt.linalg.solver._eng.lib, opt.linalg.solver._eng.ns.h
    lib.pp_profile(h, 1)
    _, iters = ip_solve_device(it, opt)
    torch.cuda.synchronize()
    ms8, l8, c8 = np.zeros(8), np.zeros(8, dtype=np.int32), np.zeros(8, dtype=np.int32)
    ms4, l4, c4 = np.zeros(4), np.zeros(4, dtype=np.int32), np.zeros(4, dtype=np.int32)
    dp, ip32 = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)
    lib.pp_phase_times(h, ms8.ctypes.data_as(dp), l8.ctypes.data_as(ip32), c8.ctypes.data_as(ip32))
    lib.pp_ip_phase_times(h, ms4.ctypes.data_as(dp), l4.ctypes.data_as(ip32), c4.ctypes.data_as(ip32))
    lib.pp_profile(h, 0)
    names = ('assemble', 'factor_levels', 'schur_tiles', 'dense_S', 'fwd_levels', 'fwd_coupling', 'coupling_solve', 'bwd_levels')
    out['phases_ms_per_iteration'] = {n: float(ms8[i]) / max(iters, 1) for i, n in enumerate(names)}
    out['phases_ms_per_iteration'].update({n: float(ms4[i]) / max(iters, 1)
                                           for i, n in enumerate(('ip_rhs', 'ip_step_lengths', 'ip_take_step', 'ip_residuals'))})
    out['phase_launches_per_iteration'] = {n: int(l8[i]) // max(iters, 1) for i, n in enumerate(names)}
    paths = (ctypes.c_int32 * 2)()
    if hasattr(lib, 'pp_bcr_block_paths') and lib.pp_bcr_block_paths(h, ctypes.cast(paths, ip32)) == 0:
        out['bcr_blocks_unpivoted_pivoted_last_factorisation'] = [int(paths[0]), int(paths[1])]
    t0 = time.time()
    ref = dq.main(HipSchurComplementLinearSolver({t: None for t in range(T)}, None, comm=SerialComm()), 0.0, 1.0, T, **args)
    out['host_producer'] = dict(wall_seconds=time.time() - t0, objective=ref.evaluate_objective())
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
