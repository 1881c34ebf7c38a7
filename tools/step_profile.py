"""Run ON THE GPU BOX: where the HOST spends its time in one bench step (do_numeric_factorization + do_back_solve on
device-resident containers, right-hand side announced) -- cProfile over N steps of a small share, so that the GPU work
is short and the host path dominates.  usage: python tools/step_profile.py [blocks] [steps]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
from parapint_amd.linalg.comm import SerialComm
from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
comm = SerialComm()
model = SyntheticKKT(blocks, 1000, 4, 200)
solver = HipSchurComplementLinearSolver({i: None for i in range(blocks)}, None, comm=comm, result_buffers=2)
dk = model.build_device_kkt(comm=comm)
solver.do_symbolic_factorization(dk)
dev = torch.device('cuda', 0)
host = np.zeros(tuple(dk.sources[0].shape))
for b, ndx in enumerate(dk.slots[0]):
    host[:, b] = model.block_sources(ndx, None)
m = dk.with_sources({0: torch.from_numpy(host).to(dev)})
rhs = solver.device_vector_from_host(model.build_rhs(comm=comm))


def step():
    solver.prefetch_forward(rhs)
    solver.do_numeric_factorization(matrix=m, raise_on_error=False)
    return solver.do_back_solve(rhs)


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print('ms per step %.4f' % (1e3 * (time.perf_counter() - t0) / steps))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
