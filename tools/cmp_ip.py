"""Diagnostic: regularisation retries of the late interior-point iterations, device loop against host loop (GPU box)."""
import logging, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_device_ip as T
from parapint_amd.linalg.comm import SerialComm
from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
from parapint_amd.examples.stochastic_qp import random_stochastic_qp
N = 256
qps, fs = random_stochastic_qp(N, seed=2)
logging.basicConfig(level=logging.DEBUG, format='%(name)s %(message)s')
for name in ('parapint_amd.algorithms.device_interior_point', 'parapint_amd.algorithms.interior_point'):
    logging.getLogger(name).setLevel(logging.DEBUG)
print('================ device loop')
it, hist, solver = T.device_loop(qps, fs)
print('refreshes', solver.pivot_order_refreshes, 'shifted', solver.diagonal_shift_refactorizations)
print('================ host loop')
hs = HipSchurComplementLinearSolver({i: None for i in range(N)}, None, comm=SerialComm())
hi, rows = T.host_loop(qps, fs, hs)
print('refreshes', hs.pivot_order_refreshes)
