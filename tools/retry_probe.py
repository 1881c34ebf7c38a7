"""One-off: cost of one inertia-correction retry at C3 -- device fast path vs the host boundary path."""
import sys, time, numpy as np, scipy.sparse as sp
sys.path.insert(0, '.')
import torch
from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
from parapint_amd.linalg.comm import SerialComm
from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
from parapint_amd.sparse.block_containers import BlockMatrix
N = 1024
model = SyntheticKKT(N, 1000, 4, 200)
comm = SerialComm()
nprim = model.n_y + model.n_q
ncon = model.block_dim - nprim
def regularised(it, dw, dc):
    kkt = model.build_kkt(comm=comm, iteration=it)
    R = BlockMatrix(N + 1, N + 1)
    sd = np.concatenate([dw * np.ones(nprim), -dc * np.ones(ncon)])
    di = np.arange(model.block_dim, dtype=np.int32)
    for i in range(N):
        K = kkt.get_block(i, i).tocoo()
        # diagonal entries appended as duplicates (explicit zeros survive; the solver sums duplicates)
        R.set_block(i, i, sp.coo_matrix((np.concatenate([K.data, sd]), (np.concatenate([K.row, di]), np.concatenate([K.col, di]))),
                                        shape=K.shape))
        R.set_block(N, i, kkt.get_block(N, i))
    R.set_block(N, N, (dw * sp.identity(model.n_theta, format='coo')).tocoo())
    return R
solver = HipSchurComplementLinearSolver({i: None for i in range(N)}, None, comm=comm)
A0 = regularised(0, 0.0, 0.0)          # explicit zero diagonals: the union pattern from the start
solver.do_symbolic_factorization(A0)
solver.do_numeric_factorization(A0)
classes = {i: np.concatenate([np.ones(nprim, dtype=np.int8), 2 * np.ones(ncon, dtype=np.int8)]) for i in range(N)}
solver.set_regularization_classes(classes)
for dw in (1e-8, 1e-6, 1e-4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    solver.refactorize_with_diagonal_shift(dw, dw, coupling_shift=dw)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    in1 = solver.get_inertia()
    R = regularised(0, dw, dw)
    t2 = time.perf_counter()
    solver.do_numeric_factorization(R)
    t3 = time.perf_counter()
    print('delta %.0e: device fast path %.2f ms, boundary path %.1f ms (+ %.0f ms to rebuild the matrix on the host), inertia equal: %s'
          % (dw, 1e3 * (t1 - t0), 1e3 * (t3 - t2), 1e3 * (t2 - t1), in1 == solver.get_inertia()))
