"""Operand loads of the gather tasks per level as scheduled, and if up to 4 consecutive rows of a panel shared the L
operands of a source column (host only, through the test interpreter's copy of the plan).  Diagnostic for DESIGN.md
section 4 / 10.

    python tools/operand_loads.py C3|C4
"""
import ctypes, sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import solver_cases as sc
from hostsim_engine import HostSimEngine
import hostsim_util as hu
from parapint_amd.linalg.comm import SerialComm
which=sys.argv[1]
if which=='C3':
    from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
    model=SyntheticKKT(2,1000,4,200); T=2
else:
    from parapint_amd.examples.performance.schur_complement.dynamic_kkt import SyntheticDynamicKKT
    T=4; model=SyntheticDynamicKKT(T,49,2,40)
kkt=model.build_kkt(comm=SerialComm(),iteration=1)
solver=sc.new_solver(lambda: HostSimEngine(), T)
solver.do_symbolic_factorization(kkt)
L=hu.lib()
sg=solver._eng.groups[min(1,len(solver._eng.groups)-1)]
st=np.zeros(13,dtype=np.int64); L.ppsim_stats(sg.h, st.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
nl=int(st[3]); out=np.zeros(3*nl,dtype=np.int64)
L.ppsim_operand_loads(sg.h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)))
o=out.reshape(nl,3)
print('level  loads_now  loads_grouped  entries  ratio')
for l in range(nl):
    if o[l,0]: print('%5d %10d %10d %8d  %.2f'%(l,o[l,0],o[l,1],o[l,2],o[l,1]/o[l,0]))
print('total', o[:,0].sum(), o[:,1].sum(), o[:,1].sum()/o[:,0].sum(), 'levels>=2', o[2:,0].sum(), o[2:,1].sum(), o[2:,1].sum()/max(1,o[2:,0].sum()))
