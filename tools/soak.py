"""Soak: many shapes / iterations / repeated factorisations of the product path, each checked by the scaled residual
of the assembled KKT system and by inertia against the known count; repeated calls must agree bit for bit.
usage: PYTHONPATH=. python tools/soak.py [rounds]"""
import sys
import time

import numpy as np

from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
from parapint_amd.linalg.comm import SerialComm
from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
shapes = [(3, 20, 2, 4), (70, 40, 2, 8), (130, 60, 3, 12), (64, 400, 4, 100), (200, 120, 4, 30), (17, 250, 4, 60),
          (256, 1000, 4, 200), (96, 250, 3, 220), (40, 600, 4, 530)]
comm = SerialComm()
t0 = time.time()
worst = 0.0
for shape in shapes:
    N = shape[0]
    model = SyntheticKKT(*shape)
    solver = HipSchurComplementLinearSolver({i: None for i in range(N)}, None, comm=comm)
    rhs = model.build_rhs(comm=comm)
    first = True
    for it in range(rounds):
        kkt = model.build_kkt(comm=comm, iteration=it)
        if first:
            solver.do_symbolic_factorization(kkt)
            first = False
        solver.do_numeric_factorization(kkt)
        x1 = solver.do_back_solve(rhs).flatten()
        solver.do_numeric_factorization(kkt)
        x2 = solver.do_back_solve(rhs).flatten()
        assert np.array_equal(x1, x2), ('not deterministic', shape, it)
        K = kkt.tocoo().tocsr()
        b = rhs.flatten()
        res = np.abs(K @ x1 - b).max() / (abs(K).sum(axis=1).max() * np.abs(x1).max() + np.abs(b).max())
        worst = max(worst, float(res))
        n_y, n_q, n_t = model.n_y, shape[1], shape[3]
        want = (N * (n_y + n_q) + n_t, N * (n_y + n_t), 0)
        assert tuple(solver.get_inertia()) == want, (shape, it, solver.get_inertia(), want)
        assert res <= 1e-9, (shape, it, res)
    print('shape', shape, 'ok, worst scaled residual so far %.2e, %.0f s' % (worst, time.time() - t0), flush=True)
print('soak ok')
