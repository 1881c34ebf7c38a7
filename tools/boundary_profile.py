"""Where the host boundary spends its time (run on the GPU box): cProfile over do_numeric_factorization + do_back_solve
with host COO blocks in and host vectors out at C3, after two untimed iterations.

    python tools/boundary_profile.py [n_blocks]
"""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT      # noqa: E402
from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver             # noqa: E402
from parapint_amd.linalg.comm import SerialComm                                                  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = SyntheticKKT(N, 1000, 4000, 200)
comm = SerialComm()
solver = HipSchurComplementLinearSolver({i: None for i in range(N)}, None, comm=comm, result_buffers=2)
kkt = model.build_kkt(comm=comm, iteration=0)
rhs = model.build_rhs(comm=comm)
solver.do_symbolic_factorization(kkt)
if os.environ.get('PP_DECLARE_CONSTANT') == '1':
    solver.declare_constant_entries(model.constant_entries())
for it in (1, 2):
    k = model.build_kkt(comm=comm, iteration=it)
    solver.do_numeric_factorization(k)
    solver.do_back_solve(rhs)
ks = [model.build_kkt(comm=comm, iteration=it) for it in (3, 4, 5, 6)]
ts = []
pr = cProfile.Profile()
for k in ks:
    t0 = time.perf_counter()
    pr.enable()
    solver.do_numeric_factorization(k, raise_on_error=False)
    t1 = time.perf_counter()
    x = solver.do_back_solve(rhs)
    pr.disable()
    ts.append((t1 - t0, time.perf_counter() - t1))
print('numeric / back_solve ms:', [(round(1e3 * a, 2), round(1e3 * b, 2)) for a, b in ts])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(35)
print(s.getvalue())
