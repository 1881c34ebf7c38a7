"""Diagnostic: where the host-boundary iteration (do_numeric_factorization + do_back_solve with host blocks in and
host vectors out) spends its time at C3.  usage: PYTHONPATH=. python tools/boundary_profile.py [blocks]"""
import cProfile
import pstats
import sys
import time

from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
from parapint_amd.linalg.comm import SerialComm
from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = SyntheticKKT(N, 1000, 4, 200)
comm = SerialComm()
solver = HipSchurComplementLinearSolver({i: None for i in range(N)}, None, comm=comm, result_buffers=2)
kkt = [model.build_kkt(comm=comm, iteration=k) for k in range(3)]
rhs = model.build_rhs(comm=comm)
solver.do_symbolic_factorization(kkt[0])
solver.do_numeric_factorization(kkt[0])
solver.do_back_solve(rhs)
t0 = time.perf_counter()
for k in (1, 2):
    solver.do_numeric_factorization(kkt[k])
    solver.do_back_solve(rhs)
print('ms per boundary iteration: %.1f' % (1e3 * (time.perf_counter() - t0) / 2))
pr = cProfile.Profile()
pr.enable()
for k in (1, 2, 0):
    solver.do_numeric_factorization(kkt[k])
    solver.do_back_solve(rhs)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
