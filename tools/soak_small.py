"""Soak of small and odd shapes (one block, one coupling variable, ragged chunk of one instance, n_theta = n_q)."""
import numpy as np
from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
from parapint_amd.linalg.comm import SerialComm
from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
comm = SerialComm()
for shape in [(1, 10, 2, 1), (1, 5, 2, 5), (2, 7, 3, 7), (65, 10, 2, 2), (129, 12, 2, 3), (64, 30, 5, 17), (5, 300, 2, 208),
              (3, 300, 2, 209), (2, 600, 2, 513), (128, 40, 3, 9), (191, 25, 2, 6), (193, 25, 4, 16), (256, 60, 3, 33),
              (320, 20, 2, 5)]:      # (odd and even chunk counts: one and two instances per lane)
    N = shape[0]
    model = SyntheticKKT(*shape)
    solver = HipSchurComplementLinearSolver({i: None for i in range(N)}, None, comm=comm)
    rhs = model.build_rhs(comm=comm)
    for it in range(2):
        kkt = model.build_kkt(comm=comm, iteration=it)
        if it == 0:
            solver.do_symbolic_factorization(kkt)
        solver.do_numeric_factorization(kkt)
        x = solver.do_back_solve(rhs).flatten()
        K = kkt.tocoo().tocsr()
        b = rhs.flatten()
        res = np.abs(K @ x - b).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())
        want = (N * (model.n_y + shape[1]) + shape[3], N * (model.n_y + shape[3]), 0)
        assert tuple(solver.get_inertia()) == want, (shape, solver.get_inertia(), want)
        assert res <= 1e-9, (shape, res)
    print('shape', shape, 'ok', '%.1e' % res, flush=True)
print('small soak ok')
