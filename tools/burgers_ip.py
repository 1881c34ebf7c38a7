"""The interior-point loop of the Burgers control problem (parapint_amd/examples/burgers.py) over the HIP solver with the host
producer: the literal shape of BASELINE.json configs[3] is `python tools/burgers_ip.py 512 50 40` (512 time blocks x 4018
variables, 49 states between them, coupling block 50 078).  Prints iterations, wall time, the solver's share (numeric
factorisation + back-solve through host containers, new values at every iteration) and the host producer's.
python tools/burgers_ip.py [time blocks] [nfe_x] [time steps per block]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class Timer(object):
    """The start / stop labels ip_solve and the solver emit (pyomo's HierarchicalTimer protocol), summed by label."""

    def __init__(self):
        self.t, self.open, self.n = {}, {}, {}

    def start(self, name):
        self.open.setdefault(name, []).append(time.perf_counter())       # (labels nest: 'factorize' inside 'factorize')
        self.n[name] = self.n.get(name, 0) + 1

    def stop(self, name):
        self.t[name] = self.t.get(name, 0.0) + time.perf_counter() - self.open[name].pop()


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    nfe_x = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    import torch
    from parapint_amd.algorithms.interior_point import IPOptions, InteriorPointStatus, ip_solve
    from parapint_amd.examples import burgers as bg
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
    t0 = time.perf_counter()
    interface = bg.BurgersInterface(start_t=0, end_t=1, num_time_blocks=T, nfe_t=T * per, nfe_x=nfe_x)
    t_build = time.perf_counter() - t0
    solver = HipSchurComplementLinearSolver({t: None for t in range(T)}, None, comm=SerialComm(), result_buffers=2)
    options = IPOptions()
    options.linalg.solver = solver
    timer = Timer()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    status = ip_solve(interface=interface, options=options, timer=timer)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    nlp0 = interface.scenario_interface(1)
    out = dict(time_blocks=T, nfe_x=nfe_x, steps_per_block=per, variables_per_block=nlp0.n_primals(),
               block_dim=nlp0.n_primals() + nlp0.n_eq_constraints() + interface.num_states,
               n_coupling=2 * interface.num_states * (T - 1), status=str(status), objective=interface.evaluate_objective(),
               model_build_seconds=t_build, wall_seconds=wall,
               iterations=timer.n.get('eval kkt', 0), pivot_order_refreshes=solver.pivot_order_refreshes,
               timer_seconds={k: round(v, 4) for k, v in sorted(timer.t.items(), key=lambda kv: -kv[1])[:14]})
    assert status == InteriorPointStatus.optimal
    # the same problem with device-resident iterates: DeviceDynamicNLPInterface + the model's functions on the device
    from parapint_amd.algorithms.device_interior_point import ip_solve_device
    best = None
    for rep in range(2):
        dev = bg.device_interface(nfe_x, T * per, T)
        opt = IPOptions()
        opt.linalg.solver = HipSchurComplementLinearSolver({t: None for t in range(T)}, None, comm=SerialComm(), result_buffers=2)
        stats, hist = {}, []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st, iters = ip_solve_device(dev, opt, stats=stats, history=hist)
        torch.cuda.synchronize()
        cur = dict(status=str(st), iterations=iters, loop_seconds=stats['loop_s'], setup_seconds=stats['setup_s'],
                   ms_per_iteration=1e3 * stats['loop_s'] / max(iters, 1), it_per_s=iters / stats['loop_s'],
                   wall_seconds=time.perf_counter() - t0, objective=dev.evaluate_objective(),
                   final_infeasibilities=list(hist[-1][:3]), torch_ops_of_the_model_and_loop=stats['torch_ops'])
        if best is None or cur['it_per_s'] > best['it_per_s']:
            best = cur
    out['device_producer'] = best
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
