"""End-to-end fuzz of the solver class's HOST logic on the test interpreter (tests/hostsim_engine.py; no GPU): random
block-bordered KKT systems -- one to three pattern groups, uniform or mapped coupling rows, a coupling block or none --
factorised six times each with the values handed over in different ways (new COO blocks, shuffled entry orders, flat value
vectors as a dictionary / one array, constant entries declared and withdrawn, entries outside the planned pattern); every
solve is checked against dense algebra (scaled residual <= 1e-9, inertia from the eigenvalues).  Systems that ARE singular
(or have a singular diagonal block, which the reference's block factorisation cannot take either) must be reported so.

    python tools/fuzz_solver.py FIRST_SEED COUNT [PROCESSES] [--hard]     (20 000 seeds: 2 minutes on two cores)

--hard: zero Hessian entries (2 x 2 pivots) and Jacobian entries scaled by up to 1e-7 per instance and iteration (pivot
sequences that differ between the instances of a group: refreshes, variants); the scaled residual must still be <= 2e-8
whatever the condition number (the a-posteriori check of the solver class: solution_check.py) -- a solve the class cannot make
accurate has to end in its RuntimeError, which is listed as 'EXC'.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import scipy.sparse as sp
from scipy.sparse import coo_matrix


def make_pattern(rng, n_x, n_c):
    J = sp.random(n_c, n_x, density=float(rng.choice([0.1, 0.25, 0.5])), random_state=int(rng.integers(1 << 30)),
                  data_rvs=lambda k: rng.normal(size=k)).tocoo()
    J = (J + 2.0 * sp.eye(n_c, n_x)).tocoo()
    n = n_x + n_c
    rows = np.concatenate([np.arange(n_x), n_x + J.row, J.col, n_x + np.arange(n_c)])
    cols = np.concatenate([np.arange(n_x), J.col, n_x + J.row, n_x + np.arange(n_c)])
    perm = rng.permutation(rows.size)
    return dict(n_x=n_x, n_c=n_c, n=n, J=J, rows=rows[perm].astype(np.int32), cols=cols[perm].astype(np.int32), perm=perm)


def block_values(pat, h, jvals):
    v = np.concatenate([h, jvals, jvals, np.zeros(pat['n_c'])])
    return v[pat['perm']]


def one(seed, hard=False, engine=None, stats=None):
    rng = np.random.default_rng(seed)
    import solver_cases as sc
    from hostsim_engine import HostSimBoundaryEngine
    from parapint_amd.sparse.block_containers import BlockMatrix, BlockVector
    from parapint_amd.sparse.host_value_matrix import HostValueMatrix
    from parapint_amd.linalg.results import LinearSolverStatus
    N = int(rng.integers(2, 8))
    P = int(rng.integers(1, 4))
    nc = int(rng.integers(1, 7))
    mapped = bool(rng.integers(0, 2)) and nc >= 2
    pats = [make_pattern(rng, int(rng.integers(7, 22)), int(rng.integers(2, 7))) for _ in range(P)]
    which = [int(rng.integers(0, P)) for _ in range(N)]
    if rng.random() < 0.3:
        which = [0] * N
    # borders: per pattern one column choice; per block the coupling rows it touches
    bcols = []
    for p in pats:
        k = int(rng.integers(1, min(nc, p['n']) + 1)) if mapped else nc
        k = min(k, p['n'])
        bcols.append(rng.choice(p['n'], size=k, replace=False).astype(np.int32))
    brows = []
    for i in range(N):
        k = bcols[which[i]].size
        if mapped:
            brows.append(np.sort(rng.choice(nc, size=k, replace=False)).astype(np.int32))
        else:
            brows.append(np.arange(nc, dtype=np.int32)[:k])
    if not mapped and any(b.size != nc for b in brows):
        return None      # (a uniform layout needs n >= nc columns: skip)
    jbase = [rng.normal(size=p['J'].nnz) + np.sign(rng.normal(size=p['J'].nnz)) * 0.5 for p in pats]
    jblock = [jbase[which[i]] * (1.0 + 0.1 * rng.normal(size=jbase[which[i]].size)) for i in range(N)]
    bvals = [rng.normal(size=brows[i].size) + 1.0 for i in range(N)]
    Qd = rng.uniform(0.5, 1.5, size=nc) if rng.random() < 0.5 else None

    hzero = [rng.random(p['n_x']) < 0.3 for p in pats] if hard else None

    def values(it):
        r = np.random.default_rng(1000 * seed + it)
        hs = [r.uniform(0.5, 2.0, size=pats[which[i]]['n_x']) for i in range(N)]
        if hard:
            for i in range(N):
                hs[i][hzero[which[i]]] = 0.0
        return hs

    def jac(it, i):
        if not hard:
            return jblock[i]
        r = np.random.default_rng(77 * seed + 13 * it + i)
        j = jblock[i].copy()
        k = int(r.integers(0, 4))
        if k:
            sel = r.choice(j.size, size=min(k, j.size), replace=False)
            j[sel] *= 10.0 ** r.uniform(-7, 0, size=sel.size)
        return j

    extra = {}

    def build(it, shuffle=(), grown=False):
        kkt = BlockMatrix(N + 1, N + 1)
        hs = values(it)
        for i in range(N):
            p = pats[which[i]]
            v = block_values(p, hs[i], jac(it, i))
            rows, cols = p['rows'], p['cols']
            if grown and i in extra:
                er, ec, evl = extra[i]
                rows = np.concatenate([rows, er, ec]).astype(np.int32)
                cols = np.concatenate([p['cols'], ec, er]).astype(np.int32)
                v = np.concatenate([v, evl, evl])
            if i in shuffle:
                q = np.random.default_rng(seed + 7 * it + i).permutation(v.size)
                rows, cols, v = rows[q].copy(), cols[q].copy(), v[q]
            kkt.set_block(i, i, coo_matrix((v, (rows, cols)), shape=(p['n'], p['n'])))
            A = coo_matrix((bvals[i], (brows[i], bcols[which[i]])), shape=(nc, p['n']))
            kkt.set_block(N, i, A)
            kkt.set_block(i, N, A.transpose().tocoo())
        kkt.set_block(N, N, coo_matrix((nc, nc)) if Qd is None else coo_matrix(np.diag(Qd)))
        return kkt

    rhs = BlockVector(N + 1)
    for i in range(N):
        rhs.set_block(i, rng.normal(size=pats[which[i]]['n']))
    rhs.set_block(N, rng.normal(size=nc))
    solver = sc.new_solver(HostSimBoundaryEngine if engine is None else engine, N, result_buffers=int(rng.choice([0, 2])))
    pattern = build(0)
    try:
        solver.do_symbolic_factorization(pattern)
        declared = False
        same_nraw = len({pats[w]['rows'].size + bcols[w].size for w in which}) == 1
        for it in range(1, 7):
            form = rng.choice(['coo', 'coo', 'shuffled', 'flat_dict', 'flat_2d', 'declare', 'withdraw', 'grow'])
            grown = False
            if form == 'grow':
                grown = True
                for i in range(N):
                    if rng.random() < 0.6:
                        p = pats[which[i]]
                        k = int(rng.integers(1, 4))
                        er = rng.integers(1, p['n_x'], size=k)
                        ec = np.array([rng.integers(0, r) for r in er])
                        extra[i] = (er, ec, 0.05 * rng.normal(size=k))
                if declared:
                    solver.declare_constant_entries(None)
                    declared = False
            if form == 'declare' and not declared and not hard:
                mask = {}
                for i in range(N):
                    p = pats[which[i]]
                    cK = np.concatenate([np.zeros(p['n_x'], bool), np.ones(2 * p['J'].nnz + p['n_c'], bool)])[p['perm']]
                    mask[i] = (cK, np.ones(brows[i].size, bool))
                solver.declare_constant_entries(mask, check=bool(rng.integers(0, 2)))
                declared = True
            elif form == 'withdraw' and declared:
                solver.declare_constant_entries(None)
                declared = False
            kkt = build(it, shuffle=set(rng.choice(N, size=int(rng.integers(1, N + 1)), replace=False)) if form == 'shuffled' else (), grown=grown)
            handed = kkt
            if form in ('flat_dict', 'flat_2d'):
                vals = {}
                for i in range(N):
                    vals[i] = np.concatenate([kkt.get_block(i, i).data, kkt.get_block(N, i).data])
                if form == 'flat_2d' and same_nraw:
                    vals = np.stack([vals[i] for i in range(N)])
                handed = HostValueMatrix(pattern, vals, Q=kkt.get_block(N, N))
            res = solver.do_numeric_factorization(handed, raise_on_error=False)
            if res.status != LinearSolverStatus.successful:
                ev = np.linalg.eigvalsh(kkt.toarray())
                if res.status == LinearSolverStatus.singular and np.abs(ev).min() <= 1e-9 * np.abs(ev).max():
                    return None                       # (the random system IS singular: correctly reported)
                bev = [np.abs(np.linalg.eigvalsh(kkt.get_block(i, i).toarray())) for i in range(N)]
                if res.status == LinearSolverStatus.singular and min(b.min() / b.max() for b in bev) <= 1e-9:
                    return None                       # (a singular K_i: the block factorisation of the reference fails, too)
                return (seed, 'status', str(res.status), form, it, getattr(solver, '_last_error', None), float(np.abs(ev).min()))
            Kd = kkt.toarray()
            try:
                x = solver.do_back_solve(rhs)
            except RuntimeError as err:
                if 'back-solve inaccurate' not in str(err):
                    raise
                # (the class would not hand out the solution: fine for a numerically singular system -- the factorisation
                # found no exactly zero pivot --, a failure for a regular one)
                sv = np.linalg.svd(Kd, compute_uv=False)
                if sv.min() <= 1e-10 * sv.max():
                    return None
                # ... and for a numerically singular DIAGONAL BLOCK: the Schur-complement method -- the reference's as much as
                # this one -- forms K_i^-1 A_i^T, whatever the condition of the whole system (seed 42408 --hard on the device:
                # whole system 9e3, one block 3e15)
                for i in range(N):
                    bs = np.linalg.svd(kkt.get_block(i, i).toarray(), compute_uv=False)
                    if bs.min() <= 1e-10 * bs.max():
                        return None
                return (seed, 'refused', str(err)[:120], float(sv.max() / sv.min()), form, it)
            r = sc.scaled_residual(Kd, x.flatten(), rhs.flatten())
            ev = np.linalg.eigvalsh(Kd)
            if hard:
                cond = np.abs(ev).max() / max(np.abs(ev).min(), 1e-300)
                bcond = max(np.abs(b).max() / max(np.abs(b).min(), 1e-300)
                            for b in (np.linalg.eigvalsh(kkt.get_block(i, i).toarray()) for i in range(N)))
                # (round 6: every back-solve is checked on the residual, refined and -- failing that -- repaired, so the
                # backward error holds whatever the condition number; before, accuracy was asked for relative to it and
                # systems beyond 1e7 were skipped)
                if not r <= 2e-8:
                    return (seed, 'residual', r, cond, bcond, form, it)
            elif not r <= 1e-9:
                return (seed, 'residual', r, form, it)
            if np.abs(ev).min() <= 1e-9 * np.abs(ev).max():
                continue                              # (numerically singular: the sign of a rounding-level eigenvalue is not defined)
            inertia = (int((ev > 0).sum()), int((ev < 0).sum()), 0)
            if tuple(solver.get_inertia()) != inertia:
                return (seed, 'inertia', tuple(solver.get_inertia()), inertia, form, it)
    except Exception as e:
        import traceback
        return (seed, 'EXC', repr(e)[:300], traceback.format_exc()[-600:])
    finally:
        if stats is not None:
            for k in ('refinement_steps', 'solves_refined', 'solve_repairs', 'inaccurate_solves', 'pivot_order_refreshes', 'group_splits'):
                stats[k] = stats.get(k, 0) + int(getattr(solver, k, 0) or 0)
    return None


def one_hard(seed):
    return one(seed, hard=True)


if __name__ == '__main__':
    hard_mode = '--hard' in sys.argv
    sys.argv = [a for a in sys.argv if a != '--hard']
    from multiprocessing import Pool
    s0, n = int(sys.argv[1]), int(sys.argv[2])
    t0 = time.time()
    bad = []
    with Pool(int(sys.argv[3]) if len(sys.argv) > 3 else 2) as p:
        for i, r in enumerate(p.imap_unordered(one_hard if hard_mode else one, range(s0, s0 + n), chunksize=4)):
            if r is not None:
                bad.append(r); print(r, flush=True)
            if (i + 1) % 100 == 0:
                print('done', i + 1, 'bad', len(bad), 'elapsed %.0fs' % (time.time() - t0), flush=True)
    print('TOTAL', n, 'bad', len(bad))
