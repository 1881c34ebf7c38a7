"""A two-stage stochastic program with NONLINEAR scenario problems (the reference's stochastic interface takes any Pyomo model
per scenario, sc_ip_interface.py:1028-1060; its one example, the farmer problem, is linear): every scenario i has copies x of
the n_f first-stage variables and n_y recourse variables y,

    min  1/2 |x - a|^2 + sum_j [ exp(y_j) - b_ij y_j ]     s.t.  y_j + 0.1 y_j^3 - x_{j mod n_f} - d_ij = 0,   -1.5 <= y_j <= 1.5

given twice: as NLP objects for the host interface (``CallbackNLP``) and as a device model for
``DeviceStochasticNLPInterface`` (the same formulas over [row][lane] arrays, lane = scenario)."""
import numpy as np
from scipy.sparse import coo_matrix

from parapint_amd.interfaces.interface import CallbackNLP


class Scenario(object):
    def __init__(self, n_f, n_y, a, b, d):
        self.n_f, self.n_y = int(n_f), int(n_y)
        self.a, self.b, self.d = (np.asarray(v, dtype=np.double) for v in (a, b, d))
        self.idx = np.arange(self.n_y) % self.n_f
        j = np.arange(self.n_y)
        self.jrow, self.jcol = np.concatenate([j, j]), np.concatenate([self.n_f + j, self.idx])
        self.n = self.n_f + self.n_y

    def nlp(self):
        """The scenario as an object with the NLP protocol (host interface, pattern source of the device producer)."""
        nf, ny, n = self.n_f, self.n_y, self.n
        dg = np.arange(n)

        def hess(v, ye, yi, of):
            y = v[nf:]
            return coo_matrix((np.concatenate([np.full(nf, of), of * np.exp(y) + ye * 0.6 * y]), (dg, dg)), shape=(n, n))
        return CallbackNLP(
            x0=np.zeros(n),
            f=lambda v: 0.5 * float(np.sum((v[:nf] - self.a) ** 2)) + float(np.sum(np.exp(v[nf:]) - self.b * v[nf:])),
            grad=lambda v: np.concatenate([v[:nf] - self.a, np.exp(v[nf:]) - self.b]),
            hess_lag=hess,
            c_eq=lambda v: v[nf:] + 0.1 * v[nf:] ** 3 - v[self.idx] - self.d,
            jac_eq=lambda v: coo_matrix((np.concatenate([1.0 + 0.3 * v[nf:] ** 2, -np.ones(ny)]), (self.jrow, self.jcol)),
                                        shape=(ny, n)),
            lb=np.concatenate([np.full(nf, -np.inf), np.full(ny, -1.5)]), ub=np.concatenate([np.full(nf, np.inf), np.full(ny, 1.5)]))


def random_scenarios(n_scenarios, n_f=3, n_y=7, seed=0):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=n_f)
    out = []
    for _ in range(n_scenarios):
        sc = Scenario(n_f, n_y, a, rng.uniform(0.5, 2.0, size=n_y), rng.normal(scale=0.5, size=n_y))
        nlp = sc.nlp()
        nlp.scenario = sc
        out.append(nlp)
    return out, [np.arange(n_f)] * n_scenarios


class ScenarioDeviceModel(object):
    """The scenario functions for all lanes of a pattern group at once (lane = scenario), written against the array operations
    numpy and torch share."""

    def __init__(self, nlps, bpad):
        scs = [o.scenario for o in nlps]
        q = scs[0]
        assert all((s.n_f, s.n_y) == (q.n_f, q.n_y) for s in scs)
        self.q = q
        self._host = tuple(np.stack([getattr(s, k) for s in scs], axis=1) for k in ('a', 'b', 'd'))     # [entry][lane]
        self._consts = None

    def evaluate(self, W, src, data, layout):
        q = self.q
        nf, ny, n = q.n_f, q.n_y, q.n
        if self._consts is None:
            to = (lambda v: W.new_tensor(v)) if hasattr(W, 'new_tensor') else (lambda v: np.asarray(v, dtype=np.double))
            self._consts = tuple(to(v) for v in self._host)
        a, b, d = self._consts
        x, y = W[0:nf], W[nf:n]
        lam = W[layout['y_eq']:layout['y_eq'] + ny]
        ey = np.exp(y) if isinstance(y, np.ndarray) else y.exp()
        data[0:nf] = x - a
        data[nf:n] = ey - b
        data[n:n + ny] = -(y + 0.1 * y ** 3 - x[q.idx] - d)
        data[layout['obj_row']] = 0.5 * ((x - a) ** 2).sum(0) + (ey - b * y).sum(0)
        src[layout['jac']:layout['jac'] + ny] = 1.0 + 0.3 * y ** 2                      # (the -1 entries are constant)
        src[layout['hess'] + nf:layout['hess'] + n] = ey + lam * 0.6 * y                 # (the first-stage diagonal is constant)
