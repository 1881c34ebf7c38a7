"""Multiply-adds of the Schur update by elimination level of the contributing pivot (host only; C3 block).  Diagnostic:
could the update of the lower levels run beside the factorisation of the upper ones?  

    python tools/schur_work_by_level.py
"""
import ctypes, sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import solver_cases as sc
from hostsim_engine import HostSimEngine
import hostsim_util as hu
from parapint_amd.linalg.comm import SerialComm
from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
model=SyntheticKKT(2,1000,4,200)
kkt=model.build_kkt(comm=SerialComm(),iteration=1)
solver=sc.new_solver(lambda: HostSimEngine(), 2)
solver.do_symbolic_factorization(kkt)
L=hu.lib(); sg=solver._eng.groups[0]
st=np.zeros(13,dtype=np.int64); L.ppsim_stats(sg.h, st.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
npiv=int(st[2]); nl=int(st[3])
lev=np.zeros(npiv,dtype=np.int32); w=np.zeros(npiv,dtype=np.int32); nc=np.zeros(npiv,dtype=np.int32)
L.ppsim_get_piv_level(sg.h, lev.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), w.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
L.ppsim_get_ncrow(sg.h, nc.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
work=w*nc*(nc+1)//2
tot=work.sum()
print('schur fma total', tot)
cum=0
for l in range(nl):
    m=lev==l
    cum+=work[m].sum()
    print('level %2d: pivots %5d with coupling rows %5d, max rows %4d, fma %8d (%.1f %%), cumulative %.1f %%'%(l, m.sum(), (nc[m]>0).sum(), nc[m].max() if m.any() else 0, work[m].sum(), 100*work[m].sum()/tot, 100*cum/tot))
