"""A dynamic optimisation problem through the time-block Schur-complement interface, Pyomo-free.

The reference's dynamic examples (``examples/dynamics.py``, ``examples/burgers.py:63-176``) build Pyomo.DAE models of one
time interval and hand them to ``MPIDynamicSchurComplementInteriorPointInterface``; neither Pyomo nor ASL is available, so
the same derivation pattern is exercised with a linear-quadratic tracking problem given as a ``QuadraticProgram`` per time
block (a 1-d diffusion equation with distributed actuators, explicit Euler in time):

    time block [t0, t1], nfe steps of length dt:  states y_0..y_nfe (n_s each), controls u_0..u_{nfe-1} (n_u each)
        y_{k+1} = y_k + dt (nu * Lap y_k + B u_k)                                  equality constraints
        y_0 = y_init                     (time block 0 only: add_init_conditions)   equality constraints
        -u_max <= u_k <= u_max, y_k <= y_max                                        bounds
        -rate * dt <= u_{k+1} - u_k <= rate * dt                                    inequality constraints
        min  dt/2 sum_{k=1..nfe} |y_k - y_ref(t_k)|^2 + r dt/2 sum_k |u_k|^2

start states = y_0, end states = y_nfe; the interface ties y_nfe of a block to y_0 of the next through the coupling
states.  ``monolithic_qp`` states the same problem over the whole horizon as ONE QuadraticProgram (the check of the
decomposition in the tests)."""
import numpy as np
from scipy.sparse import coo_matrix, diags, identity

from parapint_amd.algorithms.interior_point import IPOptions, InteriorPointStatus, ip_solve
from parapint_amd.interfaces.interface import QuadraticProgram
from parapint_amd.interfaces.schur_complement.sc_ip_interface import MPIDynamicSchurComplementInteriorPointInterface


class DiffusionControl(MPIDynamicSchurComplementInteriorPointInterface):
    def __init__(self, start_t, end_t, num_time_blocks, nfe_per_block=4, n_states=8, n_controls=2, comm=None, **model):
        self._configure(nfe_per_block, n_states, n_controls, **model)
        super(DiffusionControl, self).__init__(start_t, end_t, num_time_blocks, comm=comm)

    @classmethod
    def time_blocks(cls, start_t, end_t, num_time_blocks, local=None, **problem_args):
        """[(QuadraticProgram, start states, end states) or None] over the time blocks: what the interface asks
        build_model_for_time_block for, without the interface (the device-resident producer takes this list); local: the
        blocks to build (default all)."""
        self = cls.__new__(cls)
        self._configure(**problem_args)
        T, dt = int(num_time_blocks), (end_t - start_t) / num_time_blocks
        keep = set(range(T)) if local is None else set(local)
        return [self.build_model_for_time_block(t, dt * t, dt * (t + 1), t == 0) if t in keep else None for t in range(T)]

    def _configure(self, nfe_per_block=4, n_states=8, n_controls=2, u_max=1.5, y_max=0.9, rate=4.0, r=1e-2, nu=0.05,
                   duplicate_constraint=False):
        self.nfe, self.n_s, self.n_u = int(nfe_per_block), int(n_states), int(n_controls)
        self.u_max, self.y_max, self.rate, self.r, self.nu = u_max, y_max, rate, r, nu
        # the last dynamics equation of every time block stated twice: a rank-deficient Jacobian, the KKT matrix is
        # singular at every iterate and the inertia-correction loop (interior_point.py:364-392) has to regularise it
        self.duplicate_constraint = bool(duplicate_constraint)
        h = 1.0 / (self.n_s + 1)
        self.Lap = diags([np.ones(self.n_s - 1), -2 * np.ones(self.n_s), np.ones(self.n_s - 1)], [-1, 0, 1]).tocoo() / h ** 2 \
            if self.n_s > 1 else coo_matrix(np.array([[-2.0 / h ** 2]]))
        B = np.zeros((self.n_s, self.n_u))                     # actuators: hat functions over equal parts of the rod
        centres = (np.arange(self.n_u) + 0.5) / self.n_u
        grid = (np.arange(self.n_s) + 1) * h
        for j, c in enumerate(centres):
            B[:, j] = np.maximum(0.0, 1.0 - np.abs(grid - c) * self.n_u)
        self.B = coo_matrix(B)
        self.grid = grid
        self.y_init = 0.2 * np.sin(np.pi * grid)

    # ---- indices inside a time block
    def ys(self, k):
        return k * self.n_s + np.arange(self.n_s)

    def us(self, k):
        return (self.nfe + 1) * self.n_s + k * self.n_u + np.arange(self.n_u)

    def reference(self, t):
        """The profile to track: rises above the state bound for part of the horizon (active bounds)."""
        return (0.4 + 0.7 * np.sin(2.0 * np.pi * t)) * np.sin(np.pi * self.grid)

    def build_model_for_time_block(self, ndx, start_t, end_t, add_init_conditions):
        nfe, n_s, n_u = self.nfe, self.n_s, self.n_u
        dt = (end_t - start_t) / nfe
        n = n_s * (nfe + 1) + n_u * nfe
        step = (identity(n_s) + dt * self.nu * self.Lap).tocoo()
        rows, cols, vals, b = [], [], [], []
        for k in range(nfe):                                   # y_{k+1} - (I + dt nu Lap) y_k - dt B u_k = 0
            r0 = k * n_s
            rows += [r0 + np.arange(n_s), r0 + step.row, r0 + self.B.row]
            cols += [self.ys(k + 1), self.ys(k)[step.col], self.us(k)[self.B.col]]
            vals += [np.ones(n_s), -step.data, -dt * self.B.data]
        b.append(np.zeros(n_s * nfe))
        me = n_s * nfe
        if self.duplicate_constraint:
            last = me - 1
            for rr, cc, vv in list(zip(rows, cols, vals))[-3:]:
                pick = np.asarray(rr) == last
                rows.append(np.full(int(pick.sum()), me))
                cols.append(np.asarray(cc)[pick])
                vals.append(np.asarray(vv)[pick])
            b.append(np.zeros(1))
            me += 1
        if add_init_conditions:
            rows.append(me + np.arange(n_s))
            cols.append(self.ys(0))
            vals.append(np.ones(n_s))
            b.append(self.y_init)
            me += n_s
        A_eq = coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(me, n))
        rows, cols, vals = [], [], []
        for k in range(nfe - 1):                               # u_{k+1} - u_k
            r0 = k * n_u
            rows += [r0 + np.arange(n_u), r0 + np.arange(n_u)]
            cols += [self.us(k + 1), self.us(k)]
            vals += [np.ones(n_u), -np.ones(n_u)]
        mi = n_u * (nfe - 1)
        A_ineq = coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(mi, n)) \
            if mi else None
        hd, c, c0 = np.zeros(n), np.zeros(n), 0.0
        for k in range(1, nfe + 1):
            ref = self.reference(start_t + k * dt)
            hd[self.ys(k)] = dt
            c[self.ys(k)] = -dt * ref
            c0 += 0.5 * dt * float(ref @ ref)
        for k in range(nfe):
            hd[self.us(k)] = self.r * dt
        lb, ub = np.full(n, -np.inf), np.full(n, np.inf)
        for k in range(nfe):
            lb[self.us(k)], ub[self.us(k)] = -self.u_max, self.u_max
        for k in range(1, nfe + 1):
            ub[self.ys(k)] = self.y_max
        qp = QuadraticProgram(c=c, A_eq=A_eq, b_eq=np.concatenate(b), A_ineq=A_ineq,
                              ineq_lb=None if not mi else np.full(mi, -self.rate * dt),
                              ineq_ub=None if not mi else np.full(mi, self.rate * dt),
                              lb=lb, ub=ub, H=coo_matrix((hd, (np.arange(n), np.arange(n))), shape=(n, n)), c0=c0)
        return qp, self.ys(0), self.ys(nfe)


def monolithic_qp(problem_args, start_t, end_t, num_time_blocks):
    """The same problem as one QuadraticProgram: the time blocks side by side, y_0 of block t + 1 tied to y_nfe of block
    t by equality constraints (no coupling variables).  Returns (qp, offsets of the blocks' variables)."""
    class _Blocks(DiffusionControl):
        def __init__(self, *a, **k):
            self.qps = {}
            super(_Blocks, self).__init__(*a, **k)

        def build_model_for_time_block(self, ndx, start_t, end_t, add_init_conditions):
            out = super(_Blocks, self).build_model_for_time_block(ndx, start_t, end_t, add_init_conditions)
            self.qps[ndx] = out
            return out
    blocks = _Blocks(start_t, end_t, num_time_blocks, **problem_args)
    T = num_time_blocks
    qps = [blocks.qps[t][0] for t in range(T)]
    off = np.concatenate([[0], np.cumsum([q.n for q in qps])])
    n = int(off[-1])

    def shifted(mats, roff):
        rows = np.concatenate([m.row + r for m, r in zip(mats, roff)])
        cols = np.concatenate([m.col + o for m, o in zip(mats, off[:-1])])
        return rows, cols, np.concatenate([m.data for m in mats])
    eoff = np.concatenate([[0], np.cumsum([q.A_eq.shape[0] for q in qps])])
    ioff = np.concatenate([[0], np.cumsum([q.A_ineq.shape[0] for q in qps])])
    er, ec, ev = shifted([q.A_eq for q in qps], eoff[:-1])
    n_s = blocks.n_s
    lr, lc, lv = [], [], []
    for t in range(T - 1):                                     # y_0 of block t + 1 - y_nfe of block t = 0
        r = eoff[-1] + t * n_s + np.arange(n_s)
        lr += [r, r]
        lc += [off[t + 1] + blocks.qps[t + 1][1], off[t] + blocks.qps[t][2]]
        lv += [np.ones(n_s), -np.ones(n_s)]
    me = int(eoff[-1]) + n_s * (T - 1)
    A_eq = coo_matrix((np.concatenate([ev] + lv), (np.concatenate([er] + lr), np.concatenate([ec] + lc))), shape=(me, n))
    ir, ic, iv = shifted([q.A_ineq for q in qps], ioff[:-1])
    hr, hc, hv = shifted([q.H for q in qps], off[:-1])
    cat = lambda name: np.concatenate([getattr(q, name) for q in qps])
    qp = QuadraticProgram(c=cat('c'), A_eq=A_eq, b_eq=np.concatenate([cat('b_eq'), np.zeros(n_s * (T - 1))]),
                          A_ineq=coo_matrix((iv, (ir, ic)), shape=(int(ioff[-1]), n)), ineq_lb=cat('ineq_lb'),
                          ineq_ub=cat('ineq_ub'), lb=cat('lb'), ub=cat('ub'), H=coo_matrix((hv, (hr, hc)), shape=(n, n)),
                          c0=sum(q.c0 for q in qps))
    return qp, off


def main(linear_solver, start_t=0.0, end_t=1.0, num_time_blocks=4, comm=None, **problem_args):
    """dynamics.py / burgers.py: build the interface, solve, return it (the solution is in its time blocks)."""
    interface = DiffusionControl(start_t, end_t, num_time_blocks, comm=comm, **problem_args)
    options = IPOptions()
    options.linalg.solver = linear_solver
    status = ip_solve(interface=interface, options=options)
    assert status == InteriorPointStatus.optimal
    return interface
