"""Random plan fuzz on the test interpreter (tests/hostsim: the C++ mirror of the kernels' arithmetic over the product's own
symbolic code; no GPU): time blocks of the dynamic problem, random saddle-point blocks and synthetic scenario blocks under
random PlanOptions (chain fronts, tile tasks, front sweeps, their size limits, batch and mapping hints), each checked against
dense algebra (S, inertia, both sweeps: tests/test_symbolic_hostsim.py:check_block).  A block that fails is tried again under
the plan without the round-5 features: only a block that passes there counts as FAIL ("hard" = neither plan solves it to
1e-8).  Random blocks with a condition number beyond 1e10 are skipped (two thirds of the random saddle blocks are singular).

    python tools/fuzz_plans.py FIRST_SEED COUNT        (six processes; round 5: 4300 blocks, no FAIL, no exception)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np   # noqa: E402

def one(seed):
    rng = np.random.default_rng(seed)
    import test_symbolic_hostsim as ts
    import hostsim_util as hu
    L = hu.lib()
    kind = rng.integers(0, 3)
    opts = {
        'chain_wmax': int(rng.choice([4, 8, 16, 32, 64])),
        'chain_min_panels': int(rng.choice([2, 3, 5])),
        'chain_min_rows': int(rng.choice([4, 8, 16, 32])),
        'chain_lds_doubles': int(rng.choice([400, 1500, 4000, 9000])),
        'chain_sweeps': int(rng.integers(0, 2)),
        'chain_tiles': int(rng.integers(0, 2)),
        'tile_task_records': int(rng.choice([4, 8, 24, 48])),
        'tile_min_entries': int(rng.choice([8, 64, 256])),
        'tile_panels': int(rng.integers(0, 2)),
        'chain_fronts': int(rng.random() < 0.9),
    }
    tune = ','.join('%s=%d' % kv for kv in opts.items())
    os.environ['PP_PLAN_TUNE'] = tune
    batch = int(rng.choice([0, 1, 6, 70, 510]))
    mapped = int(rng.integers(0, 2))
    L.ppsim_set_batch_hint(batch)
    L.ppsim_set_mapped_hint(mapped)
    desc = None
    try:
        if kind == 0:
            T = int(rng.integers(3, 9)); n_s = int(rng.integers(2, 40)); n_u = int(rng.integers(1, 4)); nfe = int(rng.integers(2, 12))
            t = int(rng.choice([0, 1, T - 1]))
            desc = ('time_block', T, n_s, n_u, nfe, t)
            K, A = ts._time_block(T, n_s, n_u, nfe, t)
        elif kind == 1:
            n_x = int(rng.integers(10, 120)); n_c = int(rng.integers(2, max(3, n_x // 2))); nb = int(rng.integers(1, 12))
            desc = ('saddle', n_x, n_c, nb, seed)
            K, A = ts.random_saddle(n_x, n_c, min(nb, n_x + n_c), seed, zero_h_frac=float(rng.choice([0.0, 0.2, 0.3])),
                                    density=float(rng.choice([0.05, 0.15, 0.3])))
        else:
            from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
            shp = (2, int(rng.integers(5, 60)), int(rng.integers(2, 5)), int(rng.integers(1, 10)))
            shp = (shp[0], shp[1], shp[2], min(shp[3], shp[1]))
            desc = ('synthetic',) + shp
            m = SyntheticKKT(*shp)
            K, A = m.block_matrix(0, 3), m.border_matrix()
        ev = np.linalg.eigvalsh(ts.sym_dense(K))
        if np.abs(ev).min() <= 1e-10 * np.abs(ev).max():
            return None                       # (the random block is singular: nothing to check)
        try:
            ts.check_block(K, A, rtol=1e-8)
        except AssertionError:
            # a numerically hard random block? compare with the plan without the new features before calling it a bug
            os.environ['PP_PLAN_TUNE'] = 'chain_fronts=0,chain_tiles=0'
            try:
                ts.check_block(K, A, rtol=1e-8)
                base_ok = True
            except AssertionError:
                base_ok = False
            return (seed, desc, tune, batch, mapped, 'FAIL' if base_ok else 'hard')
        return None
    except Exception as e:
        return (seed, desc, tune, batch, mapped, 'EXC ' + repr(e)[:200])
    finally:
        L.ppsim_set_batch_hint(0); L.ppsim_set_mapped_hint(0)

if __name__ == '__main__':
    from multiprocessing import Pool
    s0, n = int(sys.argv[1]), int(sys.argv[2])
    t0 = time.time()
    bad = []
    with Pool(6) as p:
        for i, r in enumerate(p.imap_unordered(one, range(s0, s0 + n), chunksize=4)):
            if r is not None:
                bad.append(r); print(r, flush=True)
            if (i + 1) % 200 == 0:
                print('done', i + 1, 'bad', len(bad), 'elapsed %.0fs' % (time.time() - t0), flush=True)
    print('TOTAL', n, 'bad', len(bad))
