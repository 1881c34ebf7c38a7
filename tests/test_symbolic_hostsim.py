"""Host-side symbolic plan + task schedule, executed by the TEST-ONLY interpreter
(tests/hostsim) and checked against dense linear algebra.  No GPU needed: this validates
ordering, static 1x1/2x2 pivots, panel structure, update runs, Schur tiles and the
forward/backward schedules that the HIP kernels execute verbatim."""
import numpy as np
import pytest
import scipy.sparse as sp

from hostsim_util import HostSim, bk_factor_solve
from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT


def sym_dense(K):
    Kd = sp.coo_matrix(K).toarray()
    return np.tril(Kd) + np.tril(Kd, -1).T


def check_block(K, A, rtol=1e-9, expect_2x2=None):
    hs = HostSim(K, A)
    rc, S, inertia = hs.factor()
    Kd = sym_dense(K)
    Ad = sp.coo_matrix(A).toarray()
    n, nc = hs.n, hs.nc
    assert rc == 0
    Sref = -Ad @ np.linalg.solve(Kd, Ad.T)
    scale = max(1.0, np.abs(Sref).max())
    assert np.abs(S - Sref).max() <= rtol * scale
    ev = np.linalg.eigvalsh(Kd)
    assert inertia == (int((ev > 0).sum()), int((ev < 0).sum()), 0)
    rng = np.random.default_rng(5)
    r = rng.normal(size=n)
    W = hs.forward(r)
    rc_ref = -Ad @ np.linalg.solve(Kd, r)
    assert np.abs(W[n:] - rc_ref).max() <= rtol * max(1.0, np.abs(rc_ref).max())
    xc = rng.normal(size=nc)
    x = hs.backward(W, xc)
    xref = np.linalg.solve(Kd, r - Ad.T @ xc)
    assert np.abs(x - xref).max() <= rtol * max(1.0, np.abs(xref).max())
    if expect_2x2 is not None:
        assert (hs.stats['n_2x2'] > 0) == expect_2x2
    return hs


@pytest.mark.parametrize('shape', [(3, 20, 2, 4), (4, 50, 3, 6), (2, 120, 4, 30)])
def test_synthetic_blocks(shape):
    m = SyntheticKKT(*shape)
    hs = check_block(m.block_matrix(0), m.border_matrix())
    assert hs.stats['n_levels'] < hs.stats['npiv'] // 4      # shallow schedule, not a chain


def test_synthetic_c3_shape_plan_is_shallow_and_sparse():
    m = SyntheticKKT(1, 1000, 4, 200)
    hs = HostSim(m.block_matrix(0), m.border_matrix())
    st = hs.stats
    assert st['n'] == 9200 and st['nc'] == 200
    assert st['n_levels'] <= 24
    assert st["nnz_L"] <= 1.75 * 20192         # nnz(tril K_i) = 20 192 (supernode padding included)
    rc, S, inertia = hs.factor()
    assert rc == 0 and inertia == (5000, 4200, 0)


def random_saddle(n_x, n_c, n_border, seed, zero_h_frac=0.3, density=0.15):
    rng = np.random.default_rng(seed)
    h = rng.uniform(0.5, 2.0, size=n_x)
    h[rng.random(n_x) < zero_h_frac] = 0.0
    H = sp.diags(h) + 0.0 * sp.eye(n_x)
    J = sp.random(n_c, n_x, density=density, random_state=seed, data_rvs=lambda k: rng.normal(size=k)).tocsr()
    # make J full row rank w.h.p. by adding a shifted identity part
    J = J + sp.eye(n_c, n_x, k=0) * 2.0
    K = sp.bmat([[H, J.T], [J, None]]).tocoo()
    K = (K + sp.coo_matrix(([0.0] * (n_x + n_c), (range(n_x + n_c), range(n_x + n_c))), shape=K.shape)).tocoo()
    cols = rng.choice(n_x + n_c, size=n_border, replace=False)
    A = sp.coo_matrix((-np.ones(n_border), (np.arange(n_border), cols)), shape=(n_border, n_x + n_c))
    return K, A


@pytest.mark.parametrize('seed,frac', [(0, 0.3), (4, 0.3), (5, 0.3), (7, 0.3), (1, 0.2), (6, 0.2)])
def test_random_saddle_point_blocks(seed, frac):
    K, A = random_saddle(40, 15, 6, seed, zero_h_frac=frac)
    assert np.linalg.cond(sym_dense(K)) < 1e10
    check_block(K, A, rtol=1e-7)


@pytest.mark.parametrize('seed,frac', [(2, 0.3), (3, 0.3), (1, 0.3)])
def test_singular_saddle_point_blocks_are_flagged(seed, frac):
    K, A = random_saddle(40, 15, 6, seed, zero_h_frac=frac)
    assert np.linalg.cond(sym_dense(K)) > 1e14          # structurally rank deficient draw
    hs = HostSim(K, A)
    rc, S, inertia = hs.factor()
    assert rc == 2 and inertia[2] >= 1


def test_forced_two_by_two_pivots():
    # [[0, B], [B^T, 0]] has no usable 1x1 pivot anywhere
    rng = np.random.default_rng(1)
    B = np.diag(rng.uniform(1, 2, size=6)) + np.diag(rng.uniform(0.1, 0.3, size=5), 1)
    K = sp.bmat([[None, sp.coo_matrix(B)], [sp.coo_matrix(B.T), None]]).tocoo()
    A = sp.coo_matrix(([-1.0, -1.0], ([0, 1], [2, 9])), shape=(2, 12))
    check_block(K, A, expect_2x2=True)


def test_dense_coupling_rows():
    # every border row touches several block columns with non-unit values
    rng = np.random.default_rng(3)
    n = 30
    M = rng.normal(size=(n, n))
    K = sp.coo_matrix(M @ M.T + n * np.eye(n))
    A = sp.random(9, n, density=0.4, random_state=2, data_rvs=lambda k: rng.normal(size=k))
    check_block(K, A)


def test_chunked_panels_small_accumulator():
    m = SyntheticKKT(2, 60, 3, 24)
    K, A = m.block_matrix(0), m.border_matrix()
    hs = HostSim(K, A, acc_doubles=6)          # max 6 entries per task: forces many row chunks per panel
    rc, S, inertia = hs.factor()
    Kd = sym_dense(K)
    Ad = A.toarray()
    assert np.allclose(S, -Ad @ np.linalg.solve(Kd, Ad.T), rtol=1e-9, atol=1e-9)
    assert hs.stats['ntasks'] > hs.stats['npiv']


def test_singular_block_is_flagged():
    K = sp.coo_matrix(np.array([[1.0, 1.0, 0], [1.0, 1.0, 0], [0, 0, 2.0]]))
    A = sp.coo_matrix(([-1.0], ([0], [2])), shape=(1, 3))
    hs = HostSim(K, A)
    rc, S, inertia = hs.factor()
    assert rc == 2 and inertia[2] >= 1


@pytest.mark.parametrize('n,kind', [(1, 'spd'), (2, 'indef'), (7, 'indef'), (40, 'indef'), (40, 'spd'),
                                    (33, 'zero_diag'), (64, 'kkt')])
def test_dense_bunch_kaufman(n, kind):
    rng = np.random.default_rng(n)
    M = rng.normal(size=(n, n))
    if kind == 'spd':
        S = M @ M.T + n * np.eye(n)
    elif kind == 'indef':
        S = M + M.T
    elif kind == 'zero_diag':
        S = M + M.T
        np.fill_diagonal(S, 0.0)
    else:
        h = n // 2
        S = np.zeros((n, n))
        S[:h, :h] = np.diag(rng.uniform(1, 2, size=h))
        S[h:, :h] = rng.normal(size=(n - h, h))
        S[:h, h:] = S[h:, :h].T
    b = rng.normal(size=n)
    x, inertia = bk_factor_solve(S, b)
    assert np.allclose(S @ x, b, rtol=1e-9, atol=1e-9)
    ev = np.linalg.eigvalsh(S)
    assert inertia == (int((ev > 0).sum()), int((ev < 0).sum()), 0)


def test_dense_bunch_kaufman_singular():
    S = np.array([[1.0, 2.0, 0.0], [2.0, 4.0, 0.0], [0.0, 0.0, 0.0]])
    x, inertia = bk_factor_solve(S, np.ones(3))
    assert inertia[2] >= 1


@pytest.mark.parametrize('wmax,tol', [(2, 0), (2, 1), (4, 1), (4, 3)])
def test_supernodes_block_pivots(wmax, tol):
    """Block pivots (merged sub-pivot chains) give the same factorisation: S, inertia, solves."""
    import hostsim_util as hu
    hu.lib().ppsim_set_supernodes(wmax, tol)
    try:
        m = SyntheticKKT(2, 120, 4, 30)
        hs = check_block(m.block_matrix(0, 2), m.border_matrix())
        base_levels = hs.stats['n_levels']
        K, A = random_saddle(40, 15, 6, 0)
        check_block(K, A, rtol=1e-7)
        rng = np.random.default_rng(1)
        B = np.diag(rng.uniform(1, 2, size=6)) + np.diag(rng.uniform(0.1, 0.3, size=5), 1)
        K2 = sp.bmat([[None, sp.coo_matrix(B)], [sp.coo_matrix(B.T), None]]).tocoo()
        A2 = sp.coo_matrix(([-1.0, -1.0], ([0, 1], [2, 9])), shape=(2, 12))
        check_block(K2, A2)
    finally:
        hu.lib().ppsim_set_supernodes(0, -1)
    if tol > 0:
        hu.lib().ppsim_set_supernodes(1, 0)
        try:
            hs0 = HostSim(m.block_matrix(0, 2), m.border_matrix())
        finally:
            hu.lib().ppsim_set_supernodes(0, -1)
        assert base_levels < hs0.stats['n_levels']      # merging shortens the level schedule


# Round 3: the symbolic phase has three elimination orders (order_mode 0: one sub-pivot at a time, 1: rounds of
# independent clusters, 2: the cheaper of the two), panels with or without padding to whole block pivots
# (close_supernodes; without it updates into a block pivot of which the panel holds only some columns are single-column
# entries) and a root front (front_max > 4: the chain at the top of the tree as one wide block pivot, inverted and
# scaled by kernels of its own).  Every combination must give the same S, inertia and solves.
PLAN_VARIANTS = ['order_mode=0', 'order_mode=1', 'order_mode=0,close_supernodes=1,front_max=4',
                 'order_mode=1,close_supernodes=1', 'front_max=4', 'front_pad_frac=0.0', 'front_max=9',
                 'order_mode=1,round_relax_pop=0', 'order_mode=1,round_narrow_pop=10,round_narrow_wmax=2',
                 # round 5: tile tasks for every panel that qualifies / for every big panel whatever the ratio
                 'tile_panels=1,tile_min_entries=8', 'tile_panels=1,tile_min_entries=1,tile_load_ratio=100,tile_task_records=3']


@pytest.mark.parametrize('tune', PLAN_VARIANTS)
def test_plan_variants_give_the_same_factorisation(tune, monkeypatch):
    monkeypatch.setenv('PP_PLAN_TUNE', tune)
    for shape in [(3, 20, 2, 4), (2, 120, 4, 30)]:
        m = SyntheticKKT(*shape)
        check_block(m.block_matrix(0), m.border_matrix())
    for seed in (0, 4, 5, 7):
        K, A = random_saddle(40, 15, 6, seed)
        check_block(K, A, rtol=1e-7)
    rng = np.random.default_rng(1)
    B = np.diag(rng.uniform(1, 2, size=6)) + np.diag(rng.uniform(0.1, 0.3, size=5), 1)
    K2 = sp.bmat([[None, sp.coo_matrix(B)], [sp.coo_matrix(B.T), None]]).tocoo()
    A2 = sp.coo_matrix(([-1.0, -1.0], ([0, 1], [2, 9])), shape=(2, 12))
    check_block(K2, A2, expect_2x2=True)


def test_root_front_and_single_column_entries_are_exercised():
    """The C3-shaped block: the default plan has a root front wider than a block pivot, fewer levels and less storage
    than the round-2 plan (sequential order, padded panels, no front)."""
    import ctypes
    import hostsim_util as hu
    m = SyntheticKKT(1, 1000, 4, 200)
    hs = HostSim(m.block_matrix(0), m.border_matrix())
    L = hu.lib()
    npiv = hs.stats['npiv']
    nri = L.ppsim_nrowidx(hs.h)
    ps = np.zeros(npiv + 1, dtype=np.int32); pw = np.zeros(npiv, dtype=np.int32)
    rp = np.zeros(npiv + 1, dtype=np.int32); ri = np.zeros(nri, dtype=np.int32)
    ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))      # noqa: E731
    L.ppsim_get_struct(hs.h, ip(ps), ip(pw), ip(rp), ip(ri))
    assert pw[:-1].max() <= 4 and 4 < pw[-1] <= 15                  # only the last block pivot is wide
    assert hs.stats['n_levels'] <= 12
    assert hs.stats['usize'] <= 40000                                # round 2: 17 levels, 44 329 doubles per instance
    rc, S, inertia = hs.factor()
    assert rc == 0 and inertia == (5000, 4200, 0)
    os_env = __import__('os').environ
    os_env['PP_PLAN_TUNE'] = 'order_mode=0,close_supernodes=1,front_max=4'
    try:
        hs2 = HostSim(m.block_matrix(0), m.border_matrix())
    finally:
        del os_env['PP_PLAN_TUNE']
    assert hs2.stats['n_levels'] >= hs.stats['n_levels'] + 5 and hs2.stats['usize'] > 1.1 * hs.stats['usize']
    rc2, S2, inertia2 = hs2.factor()
    assert rc2 == 0 and inertia2 == inertia
    assert np.abs(S - S2).max() <= 1e-9 * np.abs(S2).max()


def test_invert_front_matches_dense_inverse():
    """pivot.hpp invert_front (the definition k_front_invert is tested against): static 1x1 / 2x2 sweeps on a 13 x 13
    block give its inverse and inertia."""
    import ctypes
    import hostsim_util as hu
    L = hu.lib()
    rng = np.random.default_rng(0)
    for w, sub in [(13, 0), (13, 0b0000000100101), (15, 0b10), (5, 0b1000)]:
        M = rng.normal(size=(w, w))
        A = M + M.T + np.diag(rng.uniform(3, 6, size=w) * rng.choice([-1, 1], size=w))
        for k in range(w):
            if (sub >> k) & 1:
                A[k, k] = 0.0                                       # a 2x2 sub-pivot that needs its partner
        inv = np.zeros(w * (w + 1) // 2)
        code = L.ppsim_invert_front(w, ctypes.c_uint(sub), A.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                    ctypes.c_double(0.0), ctypes.c_double(1e-13),
                                    inv.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        full = np.zeros((w, w))
        for i in range(w):
            for j in range(i + 1):
                full[i, j] = full[j, i] = inv[i * (i + 1) // 2 + j]
        assert np.abs(full @ A - np.eye(w)).max() < 1e-10
        ev = np.linalg.eigvalsh(A)
        assert (code & 15, (code >> 4) & 15, (code >> 8) & 15) == (int((ev > 0).sum()), int((ev < 0).sum()), 0)


def _time_block(T, n_s, n_u, nfe, t=1):
    from parapint_amd.examples.performance.schur_complement.dynamic_kkt import SyntheticDynamicKKT
    model = SyntheticDynamicKKT(T, n_s, n_u, nfe)
    K = model.block_matrix(t, 1).tocoo()
    A = model.border_matrix(t).tocsr()
    return K, A[np.unique(A.tocoo().row), :]


@pytest.mark.parametrize('shape', [(8, 49, 2, 4), (72, 16, 2, 8), (6, 30, 3, 10)])
@pytest.mark.parametrize('tune', ['', 'chain_wmax=8,chain_min_panels=2,chain_min_rows=8', 'chain_lds_doubles=1500'])
def test_chain_fronts_of_time_blocks(shape, tune, monkeypatch):
    """Round 5: chains of block pivots whose row sets nest exactly (the dense supernodes of a time block) are scheduled
    as ONE factor level (plan.hpp: chain fronts).  The interpreter mirrors the front kernel's sums; the result is the
    factorisation of the same matrix (S, inertia, both sweeps against dense algebra), with fewer factor levels than
    dependency levels; narrow / LDS-bounded fronts cut a chain into several."""
    import ctypes
    import hostsim_util as hu
    if tune:
        monkeypatch.setenv('PP_PLAN_TUNE', tune)
    L = hu.lib()
    L.ppsim_set_batch_hint(shape[0] - 2)
    L.ppsim_set_mapped_hint(1)
    try:
        K, A = _time_block(*shape)
        hs = HostSim(K, A)
        st = np.zeros(5, dtype=np.int32)
        L.ppsim_chain_stats(hs.h, st.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
        assert st[0] >= 1 and st[1] >= 2 and st[2] < hs.stats['n_levels']
        if 'chain_wmax=8' in tune:
            assert st[3] <= 8
        check_block(K, A)
        monkeypatch.setenv('PP_PLAN_TUNE', 'chain_fronts=0')
        hs0 = HostSim(K, A)
        L.ppsim_chain_stats(hs0.h, st.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
        assert st[0] == 0 and st[2] == hs0.stats['n_levels']
        # same storage layout, same factor to rounding
        hs.factor(); hs0.factor()
        assert hs.stats['usize'] == hs0.stats['usize']
        for a, b in ((hs.U, hs0.U), (hs.L, hs0.L), (hs.Dinv, hs0.Dinv)):
            assert np.abs(a - b).max() <= 1e-11 * max(1.0, np.abs(b).max())
    finally:
        L.ppsim_set_batch_hint(0)
        L.ppsim_set_mapped_hint(0)
