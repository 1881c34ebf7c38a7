"""GPU box: per-station 100 MHz stamps of one workgroup of k_chain_front (build variant _cst, -DPP_X_CHAINSTAMP)."""
import ctypes, os, sys, json, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ['PP_LIB_VARIANT'] = 'cst'
import torch
import solver_cases as sc
from parapint_amd import _native
solver, model = sc.case_dynamic(lambda: None, 8, 49, n_u=2, nfe=40, oracle=False)
torch.cuda.synchronize()
lib = _native.load_library()
out = (ctypes.c_ulonglong * 128)()
lib.pp_x_chain_stamps.argtypes = [ctypes.c_void_p]
assert lib.pp_x_chain_stamps(out) == 0
st = np.array(list(out), dtype=np.int64)
t0 = st[0]
print('load %.2f us' % ((st[1] - st[0]) / 100.0))
for i in range(16):
    a = st[2 + 4 * i: 6 + 4 * i]
    if a[0] == 0: break
    prev = st[1] if i == 0 else st[5 + 4 * (i - 1)]
    print('panel %2d: invert %.2f  scale %.2f  barrier %.2f  update+barrier %.2f' % (i, (a[0] - prev) / 100.0, (a[1] - a[0]) / 100.0, (a[2] - a[1]) / 100.0, (a[3] - a[2]) / 100.0))
print('total %.2f us' % ((st[100] - st[0]) / 100.0))
