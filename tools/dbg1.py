import sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, 'tests')
from scipy.sparse import coo_matrix
import torch
from parapint_amd.linalg.hip_schur_complement import HipLDLInterface, HipEngine
from hostsim_util import HostSim
mat = coo_matrix(([1, 7, 3, 7, 4, 3, 6], ([0, 0, 0, 1, 1, 2, 2], [0, 1, 2, 0, 1, 0, 2])), shape=(3, 3), dtype=np.double)
e = HipEngine()
s = HipLDLInterface(engine=e)
s.do_symbolic_factorization(mat)
r = s.do_numeric_factorization(mat, raise_on_error=False)
hs = HostSim(mat.tocsr(), sp.csr_matrix((0, 3)))
print(hs.factor()[0::2], hs.stats)
st = e.ns.group_stats(0)
print(r.status, st)
for w, nm, cnt in ((0, 'U', st['u_doubles']), (1, 'L', st['u_doubles'])):
    print(nm, 'dev ', e.get_factor(0, w, 0, cnt))
    print(nm, 'host', getattr(hs, nm))
print('Dinv dev', e.get_factor(0, 2, 0, hs.Dinv.size), 'host', hs.Dinv)
print('rawT', e.get_factor(0, 3, 0, 5))
