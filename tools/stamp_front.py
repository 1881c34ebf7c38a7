"""Where the waves of the root-front kernels (k_front_invert, k_scale_wide) spend their time (diagnostic; needs the
-DPP_X_STAMPS build: PP_LIB_VARIANT=stamps).   PP_LIB_VARIANT=stamps python tools/stamp_front.py"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT   # noqa: E402
from parapint_amd.linalg.comm import SerialComm   # noqa: E402
from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver   # noqa: E402

N = int(os.environ.get('PP_STAMP_BLOCKS', '1024'))
model = SyntheticKKT(N, 1000, 4, 200)
comm = SerialComm()
solver = HipSchurComplementLinearSolver({i: None for i in range(N)}, None, comm=comm)
dk = model.build_device_kkt(comm=comm)
solver.do_symbolic_factorization(matrix=dk)
dk.set_sources_from_host({ndx: model.block_sources(ndx, 3) for ndx in range(N)})
for _ in range(3):
    solver.do_numeric_factorization(matrix=dk)
lib, h = solver._eng.lib, solver._eng.ns.h
lib.pp_x_set_stamps.restype = ctypes.c_int
lib.pp_x_set_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
buf = torch.zeros(5120000, dtype=torch.int64, device='cuda')
torch.cuda.synchronize()
lib.pp_x_set_stamps(h, ctypes.c_void_p(buf.data_ptr()), -1)
solver.do_numeric_factorization(matrix=dk)
torch.cuda.synchronize()
lib.pp_x_set_stamps(h, None, -1)
st = buf.cpu().numpy()
for name, base in (('k_front_invert', 4000000), ('k_scale_wide', 4500000)):
    a = st[base:base + 500000].reshape(-1, 16)
    a = a[a[:, 0] > 0]
    if len(a) == 0:
        print(name, 'no stamps'); continue
    t0 = a[:, 0].min()
    print('== %s: %d waves stamped, span %.2f us' % (name, len(a), (a[:, 15].max() - t0) * 0.01))
    print('   start offsets p50 %.2f max %.2f us; lifetime p50 %.2f max %.2f us' % (
        np.median(a[:, 0] - t0) * 0.01, (a[:, 0] - t0).max() * 0.01, np.median(a[:, 15] - a[:, 0]) * 0.01, (a[:, 15] - a[:, 0]).max() * 0.01))
    prev = a[:, 0]
    for k in range(1, 16):
        cur = a[:, k]
        ok = cur > 0
        if ok.sum() == 0:
            continue
        dt = (cur[ok] - prev[ok]) * 0.01
        print('   station %2d  waves %5d  dt p50 %.2f max %.2f us' % (k, ok.sum(), np.median(dt), dt.max()))
        prev = np.where(ok, cur, prev)
