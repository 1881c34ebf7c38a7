"""A synthetic two-stage stochastic QP with one sparsity pattern for all scenarios: the
workload of the device-resident interior-point loop (``parapint_amd.algorithms.device_interior_point``).

Scenario i:   min 1/2 x' H_i x + c_i' x   s.t.  A_eq x = b_i,  lo_i <= A_in,i x <= hi_i,  lb <= x <= ub,
the first n_fs variables of every scenario are copies of the first-stage variables (tied by the nonanticipativity
constraints of parapint/interfaces/schur_complement/sc_ip_interface.py:1287-1318).  H_i is diagonally dominant with a
band of off-diagonals, the constraint matrices are sparse with scenario-dependent values; the data are built around a
common interior point so that every instance is strictly feasible.
"""
import numpy as np
from scipy.sparse import coo_matrix

from parapint_amd.interfaces.interface import QuadraticProgram


def random_stochastic_qp(n_scenarios, n=24, n_fs=4, n_eq=6, n_ineq=8, seed=0, duplicate_eq_row=False):
    """duplicate_eq_row: the last equality constraint of every scenario is stated twice -- the Jacobian loses rank and the
    KKT matrix is singular at every iterate, so the inertia-correction loop (interior_point.py:364-392) must regularise the
    constraint block in every iteration."""
    rng = np.random.default_rng(seed)
    # shared patterns
    hr = np.concatenate([np.arange(n), np.arange(1, n)])
    hc = np.concatenate([np.arange(n), np.arange(0, n - 1)])
    er = np.repeat(np.arange(n_eq), 3)
    ec = np.concatenate([np.sort(rng.choice(n, size=3, replace=False)) for _ in range(n_eq)] + [np.zeros(0, dtype=np.int64)])
    if duplicate_eq_row:
        ec[-3:] = ec[-6:-3]
    ir = np.repeat(np.arange(n_ineq), 3)
    ic = np.concatenate([np.sort(rng.choice(n, size=3, replace=False)) for _ in range(n_ineq)] + [np.zeros(0, dtype=np.int64)])
    z_star = rng.uniform(1.0, 2.0, size=n_fs)
    lb = np.zeros(n)
    ub = np.full(n, np.inf)
    ub[::3] = 6.0
    scenarios, first_stage = [], []
    for i in range(n_scenarios):
        r = np.random.default_rng(1000 * seed + i + 1)
        hd = np.concatenate([r.uniform(2.0, 4.0, size=n), r.uniform(-0.4, 0.4, size=n - 1)])
        H = coo_matrix((hd, (hr, hc)), shape=(n, n))
        ev = r.normal(size=er.size)
        if duplicate_eq_row:
            ev[-3:] = ev[-6:-3]
        Ae = coo_matrix((ev, (er, ec)), shape=(n_eq, n))
        Ai = coo_matrix((r.normal(size=ir.size), (ir, ic)), shape=(n_ineq, n))
        x_star = r.uniform(1.0, 3.0, size=n)
        x_star[:n_fs] = z_star
        b = Ae @ x_star
        mid = Ai @ x_star
        lo = mid - r.uniform(0.5, 2.0, size=n_ineq)
        hi = mid + r.uniform(0.5, 2.0, size=n_ineq)
        lo[::4] = -np.inf
        c = r.normal(size=n) / n_scenarios
        scenarios.append(QuadraticProgram(c=c, A_eq=Ae, b_eq=b, A_ineq=Ai, ineq_lb=lo, ineq_ub=hi, lb=lb, ub=ub,
                                          H=coo_matrix((hd / n_scenarios, (hr, hc)), shape=(n, n))))
        first_stage.append(np.arange(n_fs))
    return scenarios, first_stage


def c3_stochastic_qp(n_scenarios, n_q=1000, m=4, n_theta=200, seed=0, local=None, active_fraction=0.1):
    """A two-stage stochastic QP with the block structure of BASELINE.json configs[2] (the synthetic estimation problem of
    parapint/examples/performance/schur_complement/create_model.py:11-143): per scenario the n_y = m n_q measurements y and the
    n_q parameters q are the primal variables (5000 at the defaults), ``y - A_i q = 0`` the equality constraints (A_i: m
    stacked tridiagonal n_q x n_q matrices, values scenario by scenario), the objective is ``sum_k w_ik (y_k - yhat_ik)^2``
    and the first n_theta parameters are copies of the first-stage variables -- KKT blocks of dimension
    n_y + n_q + n_y + n_theta = 9200 with 200 coupling variables.  What the reference problem lacks for an interior-point
    run is added: bounds on all primal variables (two thirds of them finite; a fraction of the parameter bounds active at
    the solution).  All scenarios share their index arrays (one Jacobian structure, one object).

    local: iterable of the scenario indices to build (default: all); the other entries of the returned list are None."""
    rng = np.random.default_rng(seed)
    n_y = m * n_q
    n = n_y + n_q
    # A: m tridiagonal blocks stacked; equality rows [I | -A]
    tri_r = np.concatenate([np.arange(n_q), np.arange(1, n_q), np.arange(n_q - 1)])
    tri_c = np.concatenate([np.arange(n_q), np.arange(n_q - 1), np.arange(1, n_q)])
    ar = np.concatenate([k * n_q + tri_r for k in range(m)])
    ac = np.tile(tri_c, m)
    a_base = rng.normal(0.0, 5.0, size=ar.size)
    er = np.concatenate([np.arange(n_y), ar]).astype(np.int32)
    ec = np.concatenate([np.arange(n_y), n_y + ac]).astype(np.int32)
    hr = np.arange(n_y, dtype=np.int32)                     # Hessian: diagonal on y
    theta = rng.normal(5.0, 2.0, size=n_theta)
    # bounds of the first-stage copies are common to all scenarios
    lo_t = theta - rng.uniform(0.5, 2.0, size=n_theta)
    hi_t = theta + rng.uniform(0.5, 2.0, size=n_theta)
    fs = n_y + np.arange(n_theta)
    empty_i = coo_matrix((0, n))
    want = range(n_scenarios) if local is None else local
    scenarios = [None] * n_scenarios
    for i in want:
        r = np.random.default_rng(100003 * seed + i + 1)
        q_true = r.normal(5.0, 2.0, size=n_q)
        q_true[:n_theta] = theta
        a_i = a_base * (1.0 + 0.05 * r.standard_normal(a_base.size))
        y_true = np.zeros(n_y)
        np.add.at(y_true, ar, a_i * q_true[ac])
        yhat = y_true + 0.01 * np.abs(y_true).max() * r.standard_normal(n_y)
        w = r.uniform(1.0, 1.25, size=n_y)          # (the sum of the scenario objectives, not their mean: the blocks keep
                                                    # the scaling of the reference problem whatever the number of scenarios)
        lb, ub = np.full(n, -np.inf), np.full(n, np.inf)
        span = 4.0 * np.abs(y_true).max()
        fin = r.random(n_y) < 0.67
        lb[:n_y][fin] = (yhat - span)[fin]
        fin = r.random(n_y) < 0.67
        ub[:n_y][fin] = (yhat + span)[fin]
        lq = q_true - r.uniform(0.5, 2.0, size=n_q)
        uq = q_true + r.uniform(0.5, 2.0, size=n_q)
        act = r.random(n_q) < active_fraction               # bounds that cut the least-squares solution off
        lq[act] = q_true[act] + 0.02
        uq[act] = np.maximum(uq[act], lq[act] + 1.0)
        lq[r.random(n_q) < 0.33] = -np.inf
        uq[r.random(n_q) < 0.33] = np.inf
        lq[:n_theta], uq[:n_theta] = lo_t, hi_t
        lb[n_y:], ub[n_y:] = lq, uq
        c = np.zeros(n)
        c[:n_y] = -2.0 * w * yhat
        q = QuadraticProgram(c=c, A_eq=None, b_eq=np.zeros(n_y), A_ineq=None, lb=lb, ub=ub, H=None,
                             c0=float(np.sum(w * yhat * yhat)))
        # (blocks over the shared index arrays; H is lower triangular already)
        q.H = coo_matrix((2.0 * w, (hr, hr)), shape=(n, n))
        q.A_eq = coo_matrix((np.concatenate([np.ones(n_y), -a_i]), (er, ec)), shape=(n_y, n))
        q.A_ineq = empty_i
        scenarios[i] = q
    return scenarios, [fs] * n_scenarios
