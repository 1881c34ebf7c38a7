"""Where the device-resident interior-point loop of bench.py (ip_loop) spends its host time: cProfile of ip_solve_device
on the bench's stochastic QP (run on the GPU box).

    python tools/ip_profile.py [scenarios [symbolic pivot threshold]]
"""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parapint_amd.algorithms.device_interior_point import ip_solve_device                      # noqa: E402
from parapint_amd.algorithms.interior_point import IPOptions                                    # noqa: E402
from parapint_amd.examples.stochastic_qp import random_stochastic_qp                            # noqa: E402
from parapint_amd.interfaces.schur_complement.device_sc_ip_interface import DeviceStochasticQPInterface   # noqa: E402
from parapint_amd.linalg.comm import SerialComm                                                  # noqa: E402
from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver             # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
U_SYM = float(sys.argv[2]) if len(sys.argv) > 2 else None       # threshold of the static 1x1 / 2x2 choice (default 0.01)
qps, fsi = random_stochastic_qp(N, n=120, n_fs=10, n_eq=30, n_ineq=40, seed=7)
for rep in range(2):
    ipi = DeviceStochasticQPInterface(qps, fsi)
    ipo = IPOptions()
    ipo.linalg.solver = HipSchurComplementLinearSolver({i: None for i in range(N)}, None, comm=SerialComm(), result_buffers=2,
                                                       symbolic_pivot_threshold=U_SYM)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    if rep:
        pr.enable()
    status, iters = ip_solve_device(ipi, ipo)
    pr.disable()
    sv = ipo.linalg.solver
    print('run %d: %s, %d iterations in %.3f s; refreshes %d %s, shifted refactorizations %d' %
          (rep, status, iters, time.perf_counter() - t0, sv.pivot_order_refreshes, sv.refresh_causes,
           sv.diagonal_shift_refactorizations))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue())
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18)
print(s.getvalue())

# where a pivot-order refresh spends its time inside the library (one more run, the symbolic entry points timed)
lib = ipo.linalg.solver._eng.lib
acc = {}


def timed(name):
    fn = getattr(lib, name)

    def wrapper(*a):
        t = time.perf_counter()
        r = fn(*a)
        acc.setdefault(name, []).append(time.perf_counter() - t)
        return r
    return wrapper


ipi = DeviceStochasticQPInterface(qps, fsi)
ipo = IPOptions()
ipo.linalg.solver = HipSchurComplementLinearSolver({i: None for i in range(N)}, None, comm=SerialComm(), result_buffers=2,
                                                       symbolic_pivot_threshold=U_SYM)
lib = ipo.linalg.solver._eng.lib
saved = {}
for name in ('pp_begin_symbolic', 'pp_add_group', 'pp_end_symbolic', 'pp_set_value_map', 'pp_bind_schur_buffer', 'pp_find_zero_pivot',
             'pp_find_growth', 'pp_set_diagonal_classes'):
    saved[name] = getattr(lib, name)
    setattr(lib, name, timed(name))
ip_solve_device(ipi, ipo)
for name, fn in saved.items():
    setattr(lib, name, fn)
for name, v in acc.items():
    print('%-28s calls %4d  total %7.1f ms  mean %6.3f ms' % (name, len(v), 1e3 * sum(v), 1e3 * sum(v) / len(v)))
