"""Diagnostic: which dense factor the last factorisation used (1 = LDL^T accepted, 0 = Bunch-Kaufman fallback)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
from parapint_amd.linalg.comm import SerialComm
from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
for (N, n_q, m, n_t) in ((70, 60, 3, 24), (64, 400, 4, 100), (128, 1000, 4, 200)):
    model = SyntheticKKT(N, n_q, m, n_t)
    kkt = model.build_kkt(comm=SerialComm(), iteration=1)
    solver = HipSchurComplementLinearSolver({i: None for i in range(N)}, None, comm=SerialComm())
    solver.do_symbolic_factorization(kkt)
    solver.do_numeric_factorization(kkt)
    mode = ctypes.c_int(-1)
    solver._eng.lib.pp_get_dense_mode(solver._eng.ns.h, ctypes.byref(mode))
    print('n_c', n_t, 'dense mode', mode.value, 'inertia', solver.get_inertia())
