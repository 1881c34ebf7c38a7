"""Does the interior-point loop depend on what ran before it in the process (stale device memory)?  Runs the bench's
QP at the given scenario counts one after the other and prints iterations / refreshes / the last history row of each.

    python tools/ip_determinism.py 256 1024 256
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parapint_amd.algorithms.device_interior_point import ip_solve_device                      # noqa: E402
from parapint_amd.algorithms.interior_point import IPOptions                                    # noqa: E402
from parapint_amd.examples.stochastic_qp import random_stochastic_qp                            # noqa: E402
from parapint_amd.interfaces.schur_complement.device_sc_ip_interface import DeviceStochasticQPInterface   # noqa: E402
from parapint_amd.linalg.comm import SerialComm                                                  # noqa: E402
from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver             # noqa: E402

for arg in sys.argv[1:]:
    N = int(arg)
    qps, fsi = random_stochastic_qp(N, n=120, n_fs=10, n_eq=30, n_ineq=40, seed=7)
    ipi = DeviceStochasticQPInterface(qps, fsi)
    ipo = IPOptions()
    u_sym = float(os.environ['U_SYM']) if os.environ.get('U_SYM') else None       # threshold of the static 1x1 / 2x2 choice
    ipo.linalg.solver = HipSchurComplementLinearSolver({i: None for i in range(N)}, None, comm=SerialComm(), result_buffers=2,
                                                       symbolic_pivot_threshold=u_sym)
    if os.environ.get('ESCALATION'):
        ipo.linalg.solver.refresh_thresholds = (0.1, 0.3)
    hist = []
    try:
        status, iters = ip_solve_device(ipi, ipo, history=hist)
    except RuntimeError as e:
        status, iters = str(e), len(hist)
    sv = ipo.linalg.solver
    print(N, status, iters, 'refreshes', sv.pivot_order_refreshes, 'retries', sv.diagonal_shift_refactorizations,
          'last', hist[-1][:3] if hist else None, flush=True)
