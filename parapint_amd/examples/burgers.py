"""Optimal control of the viscous Burgers equation through the time-block Schur-complement interface, Pyomo-free.

The reference's ``examples/burgers.py:53-176`` states the problem with Pyomo.DAE (backward differences in time, central
differences in space, trapezoidal integrals) and hands one Pyomo model per time block to
``MPIDynamicSchurComplementInteriorPointInterface``.  Pyomo is not available here, so the SAME discretised problem is
written out by hand as an object with the NLP protocol the reference's ``InteriorPointInterface`` talks to (PyNumero's
``ExtendedNLP``: sizes, bounds, state, objective / constraints and their derivatives) -- a NONLINEAR problem: Hessian
and Jacobian values change at every iterate, every interior-point iteration hands the linear solver new values on the
same pattern.

    variables of a time block [t0, t1] with nt steps, interior grid points x_1 .. x_m (m = nfe_x - 1; y = u = 0 at x = 0, 1):
        y[k][i], u[k][i],  k = 0 .. nt
    pde (k = 1 .. nt):  (y[k][i] - y[k-1][i]) / dt - v (y[k][i+1] - 2 y[k][i] + y[k][i-1]) / dx^2
                        + (y[k][i+1] - y[k][i-1]) / (2 dx) * y[k][i] = r + u[k-1][i]                    (burgers.py:123-132)
    first block only:   y[0][i] = y0(x_i),  u[0][i] = 0                                                 (:105-121)
    objective:          1/2 int int (y - y0)^2 + omega u^2 dx dt  (trapezoid)  +  1/4 dx dt omega sum_i u[0][i]^2   (:140-160)

(the last term gives the controls at the start of a block their full weight: the copy of that time node at the end of the
block before drives nothing and goes to zero).  Start / end states: y[0][:] and y[nt][:] (:168-170).  The boundary values,
which the reference keeps as variables fixed by equality constraints, are eliminated."""
import math
import os

import numpy as np
from scipy.sparse import coo_matrix

from parapint_amd.algorithms.interior_point import IPOptions, InteriorPointStatus, ip_solve
from parapint_amd.interfaces.schur_complement.sc_ip_interface import MPIDynamicSchurComplementInteriorPointInterface


class BurgersNLP(object):
    def __init__(self, nfe_x, nfe_t, start_t, end_t, add_init_conditions, start_term=True, omega=0.02, v=0.01, r=0.0):
        self.m = m = int(nfe_x) - 1
        self.nt = nt = int(nfe_t)
        self.dx, self.dt = 1.0 / nfe_x, (end_t - start_t) / float(nfe_t)
        self.omega, self.v, self.r = omega, v, r
        self.init_conditions, self.start_term = bool(add_init_conditions), bool(start_term)
        x = (np.arange(m) + 1) * self.dx
        self.y0 = np.where(x <= 0.5, 1.0, 0.0)                          # _y_init_rule
        self.n = 2 * (nt + 1) * m
        self.me = nt * m + (2 * m if add_init_conditions else 0)
        self._primals = np.zeros(self.n)                               # (Pyomo variables without a value start at 0)
        self._duals_eq = np.zeros(self.me)
        self._obj_factor = 1.0
        self._cache = {}
        # trapezoid weights in time; the objective is 1/2 sum_k w_k dx sum_i [(y - y0)^2 + omega u^2] (+ start term)
        self.w = np.full(nt + 1, self.dt)
        self.w[0] = self.w[-1] = 0.5 * self.dt
        self._patterns()

    # ---- indices
    def yi(self, k):
        return k * self.m + np.arange(self.m)

    def ui(self, k):
        return (self.nt + 1) * self.m + k * self.m + np.arange(self.m)

    def start_states(self):
        return self.yi(0)

    def end_states(self):
        return self.yi(self.nt)

    def _patterns(self):
        m, nt = self.m, self.nt
        i = np.arange(m)
        rows, cols, kind = [], [], []           # Jacobian entries in a fixed order; kind names the formula of the value
        for k in range(1, nt + 1):
            r0 = (k - 1) * m
            rows += [r0 + i, r0 + i, r0 + i[:-1], r0 + i[1:], r0 + i]
            cols += [self.yi(k), self.yi(k - 1), self.yi(k)[1:], self.yi(k)[:-1], self.ui(k - 1)]
            kind += [np.full(m, 0), np.full(m, 1), np.full(m - 1, 2), np.full(m - 1, 3), np.full(m, 4)]
        if self.init_conditions:
            r0 = nt * m
            rows += [r0 + i, r0 + m + i]
            cols += [self.yi(0), self.ui(0)]
            kind += [np.full(m, 5), np.full(m, 5)]
        self.jrow, self.jcol, self.jkind = (np.concatenate(a).astype(np.int64) for a in (rows, cols, kind))
        # Hessian of the Lagrangian, lower triangle: the diagonal, then the pairs (y[k][i+1], y[k][i]) of the convective term
        d = np.arange(self.n)
        hr, hc = [d], [d]
        for k in range(1, nt + 1):
            hr.append(self.yi(k)[1:])
            hc.append(self.yi(k)[:-1])
        self.hrow, self.hcol = np.concatenate(hr).astype(np.int64), np.concatenate(hc).astype(np.int64)

    # ---- the NLP protocol
    def n_primals(self):
        return self.n

    def n_eq_constraints(self):
        return self.me

    def n_ineq_constraints(self):
        return 0

    def nnz_hessian_lag(self):
        return int(self.hrow.size)

    def nnz_jacobian_eq(self):
        return int(self.jrow.size)

    def nnz_jacobian_ineq(self):
        return 0

    def primals_lb(self):
        return np.full(self.n, -np.inf)

    def primals_ub(self):
        return np.full(self.n, np.inf)

    def ineq_lb(self):
        return np.zeros(0)

    def ineq_ub(self):
        return np.zeros(0)

    def init_primals(self):
        return np.zeros(self.n)

    def init_duals_eq(self):
        return np.zeros(self.me)

    def init_duals_ineq(self):
        return np.zeros(0)

    def set_primals(self, primals):
        self._primals = np.asarray(primals, dtype=np.double)
        self._cache = {}                    # evaluations are kept until the state changes (as PyNumero's NLP classes do)

    def get_primals(self):
        return self._primals

    def set_duals_eq(self, duals):
        self._duals_eq = np.asarray(duals, dtype=np.double)
        self._cache.pop('hess', None)

    def get_duals_eq(self):
        return self._duals_eq

    def set_duals_ineq(self, duals):
        pass

    def get_duals_ineq(self):
        return np.zeros(0)

    def set_obj_factor(self, obj_factor):
        self._obj_factor = obj_factor
        self._cache.pop('hess', None)

    def get_obj_factor(self):
        return self._obj_factor

    def _yu(self):
        m, nt = self.m, self.nt
        x = self._primals
        return x[:(nt + 1) * m].reshape(nt + 1, m), x[(nt + 1) * m:].reshape(nt + 1, m)

    def evaluate_objective(self):
        y, u = self._yu()
        f = 0.5 * self.dx * float(np.sum(self.w[:, None] * ((y - self.y0) ** 2 + self.omega * u ** 2)))
        if self.start_term:
            f += 0.25 * self.dx * self.dt * self.omega * float(np.sum(u[0] ** 2))
        return f

    def evaluate_grad_objective(self):
        y, u = self._yu()
        gy = self.dx * self.w[:, None] * (y - self.y0)
        gu = self.dx * self.w[:, None] * self.omega * u
        if self.start_term:
            gu[0] = gu[0] + 0.5 * self.dx * self.dt * self.omega * u[0]
        return np.concatenate([gy.ravel(), gu.ravel()])

    def _neighbours(self, yk):
        up, dn = np.zeros_like(yk), np.zeros_like(yk)
        up[:-1], dn[1:] = yk[1:], yk[:-1]                      # y[k][i+1], y[k][i-1] (zero at the boundary)
        return up, dn

    def evaluate_eq_constraints(self):
        if 'c' not in self._cache:
            self._cache['c'] = self._eq_constraints()
        return self._cache['c']

    def _eq_constraints(self):
        y, u = self._yu()
        out = []
        for k in range(1, self.nt + 1):
            up, dn = self._neighbours(y[k])
            out.append((y[k] - y[k - 1]) / self.dt - self.v * (up - 2.0 * y[k] + dn) / self.dx ** 2 +
                       (up - dn) / (2.0 * self.dx) * y[k] - self.r - u[k - 1])
        if self.init_conditions:
            out += [y[0] - self.y0, u[0]]
        return np.concatenate(out)

    def evaluate_ineq_constraints(self):
        return np.zeros(0)

    def evaluate_jacobian_eq(self):
        if 'jac' not in self._cache:
            self._cache['jac'] = self._jacobian_eq()
        return self._cache['jac']

    def _jacobian_eq(self):
        y, _ = self._yu()
        vals = []
        for k in range(1, self.nt + 1):
            up, dn = self._neighbours(y[k])
            vals += [1.0 / self.dt + 2.0 * self.v / self.dx ** 2 + (up - dn) / (2.0 * self.dx),      # d / d y[k][i]
                     np.full(self.m, -1.0 / self.dt),                                                # d / d y[k-1][i]
                     (-self.v / self.dx ** 2 + y[k] / (2.0 * self.dx))[:-1],                         # d / d y[k][i+1]
                     (-self.v / self.dx ** 2 - y[k] / (2.0 * self.dx))[1:],                          # d / d y[k][i-1]
                     np.full(self.m, -1.0)]                                                          # d / d u[k-1][i]
        if self.init_conditions:
            vals += [np.ones(self.m), np.ones(self.m)]
        return coo_matrix((np.concatenate(vals), (self.jrow, self.jcol)), shape=(self.me, self.n))

    def evaluate_jacobian_ineq(self):
        return coo_matrix((0, self.n))

    def evaluate_hessian_lag(self):
        if 'hess' not in self._cache:
            self._cache['hess'] = self._hessian_lag()
        return self._cache['hess']

    def _hessian_lag(self):
        m, nt = self.m, self.nt
        diag_y = self.dx * self.w[:, None] * np.ones((nt + 1, m))
        diag_u = self.dx * self.w[:, None] * self.omega * np.ones((nt + 1, m))
        if self.start_term:
            diag_u[0] = diag_u[0] + 0.5 * self.dx * self.dt * self.omega
        vals = [self._obj_factor * np.concatenate([diag_y.ravel(), diag_u.ravel()])]
        lam = self._duals_eq[:nt * m].reshape(nt, m)
        for k in range(1, nt + 1):
            # d2 c[k][i] / d y[k][i+1] d y[k][i] = 1 / (2 dx),  d2 c[k][i+1] / d y[k][i] d y[k][i+1] = -1 / (2 dx)
            vals.append((lam[k - 1][:-1] - lam[k - 1][1:]) / (2.0 * self.dx))
        return coo_matrix((np.concatenate(vals), (self.hrow, self.hcol)), shape=(self.n, self.n))


class BurgersDeviceModel(object):
    """The functions of ``BurgersNLP`` evaluated for all time blocks of a pattern group at once on [row][lane] arrays (lane =
    time block) that stay on the device: the model side of ``DeviceDynamicNLPInterface`` (what Pyomo / ASL evaluate for the
    reference at every iterate).  Written against the array operations numpy and torch share, so the same code serves the
    numpy engines of the CPU tests and device tensors; on the device the same functions run as one hand-written HIP kernel."""

    def __init__(self, nlps, bpad):
        q = nlps[0]
        for o in nlps:
            if (o.m, o.nt, o.dx, o.omega, o.v, o.r, o.init_conditions, o.start_term) != \
                    (q.m, q.nt, q.dx, q.omega, q.v, q.r, q.init_conditions, q.start_term) or not math.isclose(o.dt, q.dt, rel_tol=1e-12):
                raise ValueError('the time blocks of a pattern group must share their discretisation')
        self.q = q
        # the time step of every lane as its own NLP object computed it (blocks of equal length differ in the last bits)
        self._dt = np.array([o.dt for o in nlps], dtype=np.double).reshape(1, 1, len(nlps))
        self._w = np.stack([o.w for o in nlps], axis=1).reshape(q.nt + 1, 1, len(nlps))
        self._consts = None
        self._hip = None

    def _constants(self, like):
        if self._consts is None:
            q = self.q
            to = (lambda a: like.new_tensor(a)) if hasattr(like, 'new_tensor') else (lambda a: np.asarray(a, dtype=np.double))
            self._consts = (to(q.y0.reshape(1, q.m, 1)), to(self._w), to(self._dt))
        return self._consts

    def evaluate(self, W, src, data, layout):
        """Device tensors: the model as ONE hand-written HIP pass over the iterate (csrc/example_burgers.hip, through the
        library's C ABI on the current stream -- the solver's); numpy arrays, or PP_BURGERS_TORCH_MODEL set: the array form
        below (a hundred elementwise operations), which is also the kernel's checker."""
        if hasattr(W, 'is_cuda') and W.is_cuda and not os.environ.get('PP_BURGERS_TORCH_MODEL'):
            return self._evaluate_hip(W, src, data, layout)
        return self._evaluate(W, src, data, layout)

    def _evaluate_hip(self, W, src, data, layout):
        import ctypes
        import torch
        from parapint_amd import _native
        q = self.q
        m, nt, n, bp = q.m, q.nt, layout['n'], int(W.shape[1])
        per = 5 * m - 2
        # the kernel indexes these arrays itself: shapes are checked here, on the host
        if (n != 2 * (nt + 1) * m or bp % 64 or len(self._dt.ravel()) != bp or int(src.shape[1]) != bp or int(data.shape[1]) != bp
                or int(W.shape[0]) < layout['y_eq'] + nt * m or int(data.shape[0]) <= layout['obj_row']
                or int(data.shape[0]) < n + nt * m + (2 * m if q.init_conditions else 0)
                or int(src.shape[0]) < max(layout['jac'] + nt * per, layout['hess'] + n + nt * (m - 1))
                or not (W.is_contiguous() and src.is_contiguous() and data.is_contiguous())):
            raise ValueError('Burgers device model: arrays do not match the model')
        if self._hip is None:
            dev = W.device
            self._hip = (torch.from_numpy(np.ascontiguousarray(self._dt.ravel())).to(dev),
                         torch.from_numpy(np.ascontiguousarray(self._w.reshape(nt + 1, bp))).to(dev),
                         torch.from_numpy(np.ascontiguousarray(q.y0)).to(dev),
                         torch.zeros((nt + 1) * bp, dtype=torch.float64, device=dev), _native.load_library())
        dt, w, y0, scratch, lib = self._hip
        rc = lib.pp_example_burgers_model(ctypes.c_void_p(torch.cuda.current_stream(W.device).cuda_stream), m, nt, n, bp,
                                          int(layout['y_eq']), int(layout['hess']), int(layout['jac']), int(layout['obj_row']),
                                          1 if q.init_conditions else 0, 1 if q.start_term else 0, float(q.dx), float(q.omega),
                                          float(q.v), float(q.r), dt.data_ptr(), w.data_ptr(), y0.data_ptr(), W.data_ptr(),
                                          src.data_ptr(), data.data_ptr(), scratch.data_ptr())
        if rc != 0:
            raise RuntimeError('pp_example_burgers_model failed (%d)' % rc)

    def _evaluate(self, W, src, data, layout):
        q = self.q
        m, nt, dx, om, v, r = q.m, q.nt, q.dx, q.omega, q.v, q.r
        n, bp = layout['n'], W.shape[1]
        y0, w, dt = self._constants(W)             # (dt: [1][1][lane], w: [time node][1][lane])
        dt0 = dt[0]                                # [1][lane]
        Y = W[0:(nt + 1) * m].reshape(nt + 1, m, bp)
        U = W[(nt + 1) * m:n].reshape(nt + 1, m, bp)
        lam = W[layout['y_eq']:layout['y_eq'] + nt * m].reshape(nt, m, bp)
        # grad f
        data[0:(nt + 1) * m].reshape(nt + 1, m, bp)[...] = dx * w * (Y - y0)
        gU = data[(nt + 1) * m:n].reshape(nt + 1, m, bp)
        gU[...] = dx * w * om * U
        if q.start_term:
            gU[0] = gU[0] + 0.5 * dx * dt0 * om * U[0]
        # -c(x): the discretised equation at the time nodes 1 .. nt, then the initial conditions
        Yk = Y[1:]
        up, dn = Yk * 0.0, Yk * 0.0
        up[:, :-1], dn[:, 1:] = Yk[:, 1:], Yk[:, :-1]
        conv = (up - dn) / (2.0 * dx)
        c = (Yk - Y[:-1]) / dt - v * (up - 2.0 * Yk + dn) / dx ** 2 + conv * Yk - r - U[:-1]
        data[n:n + nt * m].reshape(nt, m, bp)[...] = -c
        if q.init_conditions:
            data[n + nt * m:n + nt * m + m] = -(Y[0] - y0[0])
            data[n + nt * m + m:n + nt * m + 2 * m] = -U[0]
        # objective value of every lane
        f = 0.5 * dx * (w * ((Y - y0) ** 2 + om * U ** 2)).sum(0).sum(0)
        if q.start_term:
            f = f + 0.25 * dx * dt0[0] * om * (U[0] ** 2).sum(0)
        data[layout['obj_row']] = f
        # Jacobian values in the entry order of BurgersNLP._patterns (per time node: d/dy[k][i], d/dy[k-1][i],
        # d/dy[k][i+1], d/dy[k][i-1], d/du[k-1][i]; the rows of the initial conditions are constant)
        per = 5 * m - 2
        J = src[layout['jac']:layout['jac'] + nt * per].reshape(nt, per, bp)
        J[:, 0:m] = 1.0 / dt + 2.0 * v / dx ** 2 + conv
        J[:, m:2 * m] = -1.0 / dt
        J[:, 2 * m:3 * m - 1] = (-v / dx ** 2 + Yk / (2.0 * dx))[:, :-1]
        J[:, 3 * m - 1:4 * m - 2] = (-v / dx ** 2 - Yk / (2.0 * dx))[:, 1:]
        J[:, 4 * m - 2:per] = -1.0
        # Hessian of the Lagrangian: the diagonal is constant (the objective's weights, written at set-up); the
        # convective term couples y[k][i+1] and y[k][i] with (lambda[k][i] - lambda[k][i+1]) / (2 dx)
        Hs = src[layout['hess'] + n:layout['hess'] + n + nt * (m - 1)].reshape(nt, m - 1, bp)
        Hs[...] = (lam[:, :-1] - lam[:, 1:]) / (2.0 * dx)


def device_interface(nfe_x, nfe_t, nblocks, comm=None, start_t=0.0, end_t=1.0):
    """``BurgersInterface`` with device-resident iterates: the same time blocks handed to ``DeviceDynamicNLPInterface``
    together with their device model."""
    from parapint_amd.interfaces.schur_complement.device_sc_ip_interface import DeviceDynamicNLPInterface
    T = int(nblocks)
    size, rank = (1, 0) if comm is None else (comm.size, comm.rank)
    per = nfe_t // T
    delta = (end_t - start_t) / T
    blocks = []
    for t in range(T):
        if t % size != rank:
            blocks.append(None)
            continue
        nlp = BurgersNLP(nfe_x, per, delta * t, delta * (t + 1), t == 0)
        blocks.append((nlp, nlp.start_states(), nlp.end_states()))
    return DeviceDynamicNLPInterface(blocks, BurgersDeviceModel, comm=comm)


class BurgersInterface(MPIDynamicSchurComplementInteriorPointInterface):
    """burgers.py:53-176: the interface of the reference's example (same constructor but for the communicator, which is
    an argument here instead of a module global)."""

    def __init__(self, start_t, end_t, num_time_blocks, nfe_t, nfe_x, comm=None):
        self.nfe_x = nfe_x
        self.dt = (end_t - start_t) / float(nfe_t)
        super(BurgersInterface, self).__init__(start_t=start_t, end_t=end_t, num_time_blocks=num_time_blocks, comm=comm)

    def build_model_for_time_block(self, ndx, start_t, end_t, add_init_conditions):
        nfe_t = math.ceil((end_t - start_t) / self.dt - 1e-9)
        nlp = BurgersNLP(self.nfe_x, nfe_t, start_t, end_t, add_init_conditions)
        return nlp, nlp.start_states(), nlp.end_states()


def main(linear_solver, nfe_x=12, nfe_t=16, nblocks=4, comm=None):
    """burgers.py:202-226 without the plots: returns the interface after a successful solve."""
    interface = BurgersInterface(start_t=0, end_t=1, num_time_blocks=nblocks, nfe_t=nfe_t, nfe_x=nfe_x, comm=comm)
    options = IPOptions()
    options.linalg.solver = linear_solver
    status = ip_solve(interface=interface, options=options)
    assert status == InteriorPointStatus.optimal
    return interface
