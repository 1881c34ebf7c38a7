"""Per-level shape of the factor schedule of one pattern group (host only, through the test interpreter's copy of the
symbolic code): how many fused / gather / scale tasks and big panels every level holds.  Diagnostic.

    python tools/plan_levels.py [n_q m n_t]      (default: the C3 block, 1000 4 200)
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT   # noqa: E402
from parapint_amd.linalg.comm import SerialComm   # noqa: E402
import solver_cases as sc   # noqa: E402
from hostsim_engine import HostSimEngine   # noqa: E402
import hostsim_util as hu   # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    n_q, m, n_t = (int(a) for a in args[:3]) if len(args) >= 3 else (1000, 4, 200)
    solver = sc.new_solver(lambda: HostSimEngine(), 2)
    if '--qp' in sys.argv:       # the KKT blocks of the C3-shaped stochastic QP (bounds on the primal variables) at its initial point
        from parapint_amd.examples.stochastic_qp import c3_stochastic_qp
        from parapint_amd.interfaces.schur_complement.device_sc_ip_interface import DeviceStochasticQPInterface
        qps, fs = c3_stochastic_qp(2, n_q=n_q, m=m, n_theta=n_t, seed=1)
        solver.do_symbolic_factorization(DeviceStochasticQPInterface(qps, fs).device_kkt_matrix().pattern)
    else:
        model = SyntheticKKT(2, n_q, m, n_t)
        solver.do_symbolic_factorization(model.build_kkt(comm=SerialComm(), iteration=0))
    print(solver.plan_stats[0])
    sg = solver._eng.groups[0]
    L = hu.lib()
    st = np.zeros(13, dtype=np.int64)
    L.ppsim_stats(sg.h, st.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    nl = int(st[3])
    out = np.zeros(8 * nl, dtype=np.int32)
    L.ppsim_level_profile(sg.h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    teams = np.zeros(4 * nl, dtype=np.int32)
    L.ppsim_level_teams(sg.h, teams.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    print('level  fused  gather  split  scale  bigpanels  rows(sum)  rows(max)  maxent | chunks/panel(max)  rows/chunk(max)  panels>4  panels>8')
    for lv, (row, tm) in enumerate(zip(out.reshape(nl, 8), teams.reshape(nl, 4))):
        print('%5d %6d %7d %6d %6d %10d %10d %10d %7d | %17d %16d %9d %9d' %
              ((lv,) + tuple(int(v) for v in row) + tuple(int(v) for v in tm)))
    solve_levels(L, sg, nl)


def solve_levels(L, sg, nl):
    out = np.zeros(4 * nl, dtype=np.int32)
    L.ppsim_solve_levels(sg.h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    print('solve schedule: level  columns  longest fwd row  fwd entries  longest column (rows)')
    for lv, row in enumerate(out.reshape(nl, 4)):
        print('%21d %8d %16d %12d %22d' % ((lv,) + tuple(int(v) for v in row)))


if __name__ == '__main__':
    main()
