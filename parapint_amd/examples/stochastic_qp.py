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
