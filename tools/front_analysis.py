"""Supernode / front statistics of a time block's plan (host only): how wide the fronts of a multifrontal formulation are."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import hostsim_util as hu
from parapint_amd.examples.performance.schur_complement.dynamic_kkt import SyntheticDynamicKKT
from scipy.sparse import coo_matrix

def main():
    n_s, n_u, nfe = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (49, 2, 40)
    model = SyntheticDynamicKKT(4, n_s, n_u, nfe)
    t = 1
    K = model.block_matrix(t, 1).tocoo()
    A = model.border_matrix(t).tocsr()
    rows = np.unique(A.tocoo().row)
    A = A[rows, :]
    L = hu.lib()
    L.ppsim_set_batch_hint(510); L.ppsim_set_mapped_hint(1)
    hs = hu.HostSim(K, A)
    print(hs.stats)
    P = hs.h
    npiv = hs.stats['npiv']; n = hs.n
    nri = L.ppsim_nrowidx(P)
    ps = np.zeros(npiv, np.int32); pw = np.zeros(npiv, np.int32); rp = np.zeros(npiv + 1, np.int32); ri = np.zeros(nri, np.int32)
    L.ppsim_get_struct(P, hu._ip(ps), hu._ip(pw), hu._ip(rp), hu._ip(ri))
    lv = np.zeros(npiv, np.int32); L.ppsim_get_levels(P, hu._ip(lv))
    piv_of_col = np.zeros(n, np.int32)
    for p in range(npiv): piv_of_col[ps[p]:ps[p] + pw[p]] = p
    parent = -np.ones(npiv, np.int64)
    rowsets = []
    for p in range(npiv):
        r = ri[rp[p]:rp[p + 1]]
        rowsets.append(set(int(x) for x in r))
        rk = r[r < n]
        if rk.size: parent[p] = piv_of_col[rk.min()]
    # greedy chain amalgamation: child -> parent if padding small
    sn_of = np.arange(npiv); cols = {p: list(range(ps[p], ps[p] + pw[p])) for p in range(npiv)}
    struct = {p: set(rowsets[p]) for p in range(npiv)}
    nchild_merged = np.zeros(npiv, np.int32)
    tol = float(os.environ.get('TOL', '0.3')); wcap = int(os.environ.get('WCAP', '64'))
    # process in order; a pivot p merges into parent's (future) supernode: do it top-down later; here bottom-up: merge parent into child's chain
    head = list(range(npiv))          # supernode representative (first pivot in chain)
    for p in range(npiv):
        q = parent[p]
        if q < 0: continue
        h = head[p]
        if nchild_merged[q]: continue
        sp = struct[h]          # structure below chain so far (rows not in chain cols)
        colsq = set(range(ps[q], ps[q] + pw[q]))
        need = (sp - colsq) | rowsets[q]
        pad = len(need) - len(rowsets[q])   # rows added to q's column(s)
        width = len(cols[h]) + pw[q]
        # padding of chain columns: rows of q not in chain struct
        padc = len(need) - len(sp - colsq)
        if width <= wcap and pad <= tol * max(8, len(rowsets[q])) and padc <= tol * max(8, len(sp)):
            cols[h] = cols[h] + list(colsq and range(ps[q], ps[q] + pw[q]))
            struct[h] = need
            head[q] = h; nchild_merged[q] = 1
            del cols[q]; del struct[q]
    sns = sorted(cols.keys())
    ws = np.array([len(cols[s]) for s in sns]); ms = np.array([len(cols[s]) + len(struct[s]) for s in sns])
    print('supernodes', len(sns), 'max width', ws.max(), 'max front', ms.max())
    big = [(int(w), int(m)) for w, m in zip(ws, ms) if m >= 32]
    print('fronts >= 32 rows:', len(big), 'sum w', sum(w for w, m in big), 'of', n)
    print('L entries dense', int((ws * ms).sum()), 'usize', hs.stats['usize'], 'flops dense', int((ws * ms * ms).sum()), 'sparse fma', hs.stats['flops_factor'])
    hist = {}
    for w, m in zip(ws, ms):
        key = (min(int(w) // 8 * 8, 64), int(m) // 16 * 16); hist[key] = hist.get(key, 0) + 1
    for k in sorted(hist): print(k, hist[k])
    # depth of supernode tree
    sn_parent = {}
    for s in sns:
        last = cols[s][-1]; q = parent[piv_of_col[last]]
        sn_parent[s] = head[q] if q >= 0 else -1
    depth = {}
    for s in reversed(sns):
        depth[s] = 0 if sn_parent[s] < 0 else depth[sn_parent[s]] + 1
    print('tree depth', max(depth.values()) + 1, 'leaves', sum(1 for s in sns if s not in set(sn_parent.values())))

main()
