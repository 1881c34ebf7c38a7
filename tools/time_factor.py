"""Time of the block factorisation alone (pp_numeric_factor_blocks, no status handling), C3 workload with device-resident
values: for A/B of kernel builds, including timing experiments whose numbers are wrong on purpose (diagnostic).

    [PP_LIB_VARIANT=name] python tools/time_factor.py [steps [blocks]]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT   # noqa: E402
from parapint_amd.linalg.comm import SerialComm   # noqa: E402
from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver   # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    N, n_q, m, n_t = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024), 1000, 4, 200
    model = SyntheticKKT(N, n_q, m, n_t)
    comm = SerialComm()
    solver = HipSchurComplementLinearSolver({i: None for i in range(N)}, None, comm=comm)
    dk = model.build_device_kkt(comm=comm)
    solver.do_symbolic_factorization(matrix=dk)
    dk.set_sources_from_host({ndx: model.block_sources(ndx, 3) for ndx in range(N)})
    solver._bind_device_matrix(dk)
    eng = solver._eng
    for _ in range(5):
        eng.numeric_factor_blocks()
    torch.cuda.synchronize()
    times = []
    for _ in range(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.numeric_factor_blocks()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    print('variant %s: factor_blocks median %.4f ms  min %.4f  (n=%d)' %
          (os.environ.get('PP_LIB_VARIANT', 'base'), float(np.median(times)), min(times), steps))


if __name__ == '__main__':
    main()
