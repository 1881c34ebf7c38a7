"""Soak at the headline size (1024 x 9200, n_c = 200): repeated factorisations of changing values, bitwise
determinism of repeated calls, residual and inertia every iteration.  usage: PYTHONPATH=. python tools/soak_c3.py [n]"""
import sys
import time
import numpy as np
from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
from parapint_amd.linalg.comm import SerialComm
from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = 1024
comm = SerialComm()
model = SyntheticKKT(N, 1000, 4, 200)
solver = HipSchurComplementLinearSolver({i: None for i in range(N)}, None, comm=comm)
rhs = model.build_rhs(comm=comm)
b = rhs.flatten()
want = (N * (model.n_y + 1000) + 200, N * (model.n_y + 200), 0)
t0 = time.time()
for it in range(rounds):
    kkt = model.build_kkt(comm=comm, iteration=it)
    if it == 0:
        solver.do_symbolic_factorization(kkt)
    xs = []
    for rep in range(3):
        solver.do_numeric_factorization(kkt)
        xs.append(solver.do_back_solve(rhs).flatten())
    assert np.array_equal(xs[0], xs[1]) and np.array_equal(xs[0], xs[2]), ('not deterministic', it)
    K = kkt.tocoo().tocsr()
    res = np.abs(K @ xs[0] - b).max() / (abs(K).sum(axis=1).max() * np.abs(xs[0]).max() + np.abs(b).max())
    assert tuple(solver.get_inertia()) == want and res <= 1e-9, (it, res, solver.get_inertia())
    print('iteration %d ok: residual %.2e, %.0f s' % (it, res, time.time() - t0), flush=True)
print('C3 soak ok')
