"""Device-resident producer of the stochastic KKT system.

The reference's interior-point interfaces rebuild the block KKT matrix and the right-hand side on the host in every
iteration (parapint/interfaces/interface.py:432-538, schur_complement/sc_ip_interface.py:1677-1710,
mpi_sc_ip_interface.py:470-478) and hand them to the linear solver as SciPy / PyNumero objects.  For a two-stage
stochastic QP only the barrier diagonals ``z_l / (x - l) + z_u / (u - x)`` (interface.py:450-465) change between
iterations; Hessian and Jacobian values are data.  This class keeps the iterates of ALL scenarios in HBM in the solver's
own [row][instance] layout and

  * writes the per-iteration KKT values straight into the source tensor of the solver's ``DeviceBlockMatrix`` (the value
    map from KKT entry to source is found once, by evaluating the host interface on tagged values),
  * assembles the right-hand side into a ``DeviceBlockVector`` (interface.py:496-538, sc_ip_interface.py:1683-1696),
  * recovers the bound-dual steps, the convergence measures and the step lengths on the device
    (interface.py:562-588, algorithms/interior_point.py:174-317, 655-758; ``parapint_amd.linalg.device_vector_ops``),
  * expresses the inertia-correction regularisation as a diagonal shift of the resident matrix
    (interface.py:590-619, sc_ip_interface.py:1736-1757 -> ``DeviceBlockMatrix.with_diagonal_shift``).

It is the producer side of the hot path, not a port of the interface classes: scenarios are given as
``QuadraticProgram`` objects with one common sparsity pattern (one pattern group), one rank.  The host class
``StochasticSchurComplementInteriorPointInterface`` stays the specification -- the value map is read off it, the initial
point comes from it, and ``tests/test_device_ip.py`` compares the iterates of both.
"""
import numpy as np

from parapint_amd.algorithms import interior_point as host_ip
from parapint_amd.interfaces.interface import QuadraticProgram
from parapint_amd.interfaces.schur_complement.sc_ip_interface import StochasticSchurComplementInteriorPointInterface
from parapint_amd.sparse.device_containers import DeviceBlockMatrix

_TAG = 1 << 22          # tags of the five source families are k * _TAG + index + 1 (exact in a double)


def _same_pattern(a, b):
    return a.shape == b.shape and np.array_equal(a.row, b.row) and np.array_equal(a.col, b.col)


class DeviceStochasticQPInterface(object):
    def __init__(self, scenarios, first_stage_indices, bounds_relaxation_factor=1e-8):
        self.scenarios = list(scenarios)
        self.N = N = len(self.scenarios)
        q0 = self.scenarios[0]
        fs0 = np.asarray(first_stage_indices[0], dtype=np.int64)
        for q, fs in zip(self.scenarios, first_stage_indices):
            if not (_same_pattern(q.H, q0.H) and _same_pattern(q.A_eq, q0.A_eq) and _same_pattern(q.A_ineq, q0.A_ineq)
                    and np.array_equal(np.asarray(fs, dtype=np.int64), fs0)):
                raise ValueError('the device producer needs scenarios with one common sparsity pattern')
        self.fs = fs0
        self.n, self.me, self.mi, self.nfs = q0.n, q0.A_eq.shape[0], q0.A_ineq.shape[0], fs0.size
        self.nb = self.n + 2 * self.mi + self.me + self.nfs            # dimension of one diagonal block
        self.nnzH, self.nnzAe, self.nnzAi = q0.H.nnz, q0.A_eq.nnz, q0.A_ineq.nnz
        # source rows: [H | A_eq | A_ineq | primal barrier diagonal | slack barrier diagonal]
        self.off = np.cumsum([0, self.nnzH, self.nnzAe, self.nnzAi, self.n, self.mi])
        self.nsrc = int(self.off[-1])
        self.host = StochasticSchurComplementInteriorPointInterface(self.scenarios, [fs0] * N)
        self.host.set_bounds_relaxation_factor(bounds_relaxation_factor)
        self._relax = bounds_relaxation_factor
        self._value_map = self._find_value_map()
        self.barrier = None
        self.solver = None

    # ------------------------------------------------------------------ value map (once, on the host)
    def _find_value_map(self):
        """(src, coef) for the COO entries of K_i followed by those of A_i, read off the HOST interface evaluated on
        tagged values: every Hessian / Jacobian / barrier-diagonal value is replaced by a number that names it."""
        q0 = self.scenarios[0]

        def tags(k, count):
            return (k * _TAG + 1 + np.arange(count)).astype(np.double)
        H = q0.H.copy(); H.data = tags(1, self.nnzH)
        Ae = q0.A_eq.copy(); Ae.data = tags(2, self.nnzAe)
        Ai = q0.A_ineq.copy(); Ai.data = tags(3, self.nnzAi)
        tq = QuadraticProgram(c=q0.c, A_eq=Ae, b_eq=q0.b_eq, A_ineq=Ai, ineq_lb=q0.ineq_lb, ineq_ub=q0.ineq_ub,
                              lb=q0.lb, ub=q0.ub, H=None)
        tq.H = H                                      # (already lower triangular; keep the entry order)
        ti = StochasticSchurComplementInteriorPointInterface([tq], [self.fs])
        nlp = ti.scenario_interface(0)
        nlp.barrier_diagonals = lambda: (tags(4, self.n), tags(5, self.mi))
        kkt = ti.evaluate_primal_dual_kkt_matrix()
        vals = np.concatenate([kkt.get_block(0, 0).tocoo().data, kkt.get_block(1, 0).tocoo().data])
        a = np.abs(vals)
        fam = (a // _TAG).astype(np.int64)
        idx = (a % _TAG).astype(np.int64) - 1
        is_src = (fam >= 1) & (fam <= 5) & (a == np.round(a)) & (idx >= 0)
        src = np.where(is_src, self.off[np.clip(fam - 1, 0, 4)] + idx, -1).astype(np.int32)
        coef = np.where(is_src, np.sign(vals), vals)
        return src, coef

    # ------------------------------------------------------------------ matrix for the symbolic phase
    def initial_state_host(self):
        """The processed initial point of ip_solve (interior_point.py:433-447, 761-799) per scenario, on the host."""
        h = self.host
        lay = {'primals': host_ip._Layout(h.init_primals(), None), 'ineq': host_ip._Layout(h.init_slacks(), None),
               'eq': host_ip._Layout(h.init_duals_eq(), None)}
        P, I, E = lay['primals'], lay['ineq'], lay['eq']
        st = dict(primals=P.flat(h.init_primals()), slacks=I.flat(h.init_slacks()), duals_eq=E.flat(h.init_duals_eq()),
                  duals_ineq=I.flat(h.init_duals_ineq()), zl=P.flat(h.init_duals_primals_lb()),
                  zu=P.flat(h.init_duals_primals_ub()), sl=I.flat(h.init_duals_slacks_lb()),
                  su=I.flat(h.init_duals_slacks_ub()))
        plb, pub = P.flat(h.primals_lb()), P.flat(h.primals_ub())
        ilb, iub = I.flat(h.ineq_lb()), I.flat(h.ineq_ub())
        host_ip.process_init(st['primals'], plb, pub)
        host_ip.process_init(st['slacks'], ilb, iub)
        host_ip.process_init_duals_lb(st['zl'], plb)
        host_ip.process_init_duals_ub(st['zu'], pub)
        host_ip.process_init_duals_lb(st['sl'], ilb)
        host_ip.process_init_duals_ub(st['su'], iub)
        st.update(plb=plb, pub=pub, ilb=ilb, iub=iub)
        return st, lay

    def device_kkt_matrix(self):
        """DeviceBlockMatrix for do_symbolic_factorization: the host KKT matrix at the processed initial point is the
        pattern (its values fix the static pivot order), the value map names the source of every entry."""
        st, lay = self.initial_state_host()
        h = self.host
        P, I, E = lay['primals'], lay['ineq'], lay['eq']
        h.set_primals(P.unflat(st['primals'])); h.set_slacks(I.unflat(st['slacks']))
        h.set_duals_eq(E.unflat(st['duals_eq'])); h.set_duals_ineq(I.unflat(st['duals_ineq']))
        h.set_duals_primals_lb(P.unflat(st['zl'])); h.set_duals_primals_ub(P.unflat(st['zu']))
        h.set_duals_slacks_lb(I.unflat(st['sl'])); h.set_duals_slacks_ub(I.unflat(st['su']))
        # One sparsity pattern for all scenarios (checked in __init__): the blocks of scenario 0 at its processed initial
        # point stand for every scenario -- the solver reads patterns and one representative value set from this
        # matrix, never the values of the others (those come from the sources on the device).  Evaluating and
        # converting all 1024 host blocks was 1.0 of the 1.3 s of the whole 1024-scenario solve.
        pattern = h.evaluate_primal_dual_kkt_matrix(only=(0,))
        K0, A0 = pattern.get_block(0, 0).tocoo(), pattern.get_block(self.N, 0).tocoo()
        pattern.set_block(0, 0, K0)
        pattern.set_block(self.N, 0, A0)
        A0t = A0.transpose().tocoo()
        pattern.set_block(0, self.N, A0t)
        for ndx in range(1, self.N):
            pattern.set_block(ndx, ndx, K0)
            pattern.set_block(self.N, ndx, A0)
            pattern.set_block(ndx, self.N, A0t)
        maps = {ndx: self._value_map for ndx in range(self.N)}
        self._init_state = st
        return DeviceBlockMatrix(pattern, maps, self.nsrc)

    # ------------------------------------------------------------------ state on the device
    def attach(self, solver, dk):
        """After do_symbolic_factorization(matrix=dk): iterates, data and bounds as [row][instance] tensors in the
        solver's lane order, constant KKT values into the source tensor, regularisation classes, right-hand side."""
        import torch
        if len(dk.slots) != 1:
            raise ValueError('the device producer handles one pattern group')
        self.solver, self.dk = solver, dk
        (gid, order), = dk.slots.items()
        self.gid, self.order = gid, list(order)
        src = dk.sources[gid]
        self.bpad, self.B = src.shape[1], len(order)
        dev = src.device
        st = self._init_state
        n, me, mi, nfs, N = self.n, self.me, self.mi, self.nfs, self.N
        lanes = np.array(self.order + [self.order[0]] * (self.bpad - self.B))      # padded lanes repeat a real scenario ...

        def dev2(per_scenario, rows):
            a = np.zeros((rows, self.bpad))
            for b, ndx in enumerate(lanes):
                a[:, b] = per_scenario(ndx)
            return torch.from_numpy(a).to(dev)

        def seg(name, size, stride):      # scenario ndx's part of a flat host vector with `stride` entries per scenario
            v = st[name]
            return lambda ndx: v[ndx * stride: ndx * stride + size]
        self.x = dev2(seg('primals', n, n), n)
        self.s = dev2(seg('slacks', mi, mi), mi)
        self.yeq = dev2(lambda ndx: st['duals_eq'][ndx * (me + nfs): ndx * (me + nfs) + me], me)
        self.ylink = dev2(lambda ndx: st['duals_eq'][ndx * (me + nfs) + me: (ndx + 1) * (me + nfs)], nfs)
        self.yin = dev2(seg('duals_ineq', mi, mi), mi)
        self.zl, self.zu = dev2(seg('zl', n, n), n), dev2(seg('zu', n, n), n)
        self.sl, self.su = dev2(seg('sl', mi, mi), mi), dev2(seg('su', mi, mi), mi)
        self.lb, self.ub = dev2(seg('plb', n, n), n), dev2(seg('pub', n, n), n)
        self.ilb, self.iub = dev2(seg('ilb', mi, mi), mi), dev2(seg('iub', mi, mi), mi)
        # ... but carry no bound and no bound dual, so that they never limit a step or enter a norm
        for t, v in ((self.lb, -np.inf), (self.ub, np.inf), (self.ilb, -np.inf), (self.iub, np.inf),
                     (self.zl, 0.0), (self.zu, 0.0), (self.sl, 0.0), (self.su, 0.0)):
            t[:, self.B:] = v
        self.z = torch.from_numpy(st['primals'][N * n:N * n + nfs].copy()).to(dev)      # coupling variables (free)
        qs = self.scenarios
        self.c = dev2(lambda ndx: qs[ndx].c, n)
        self.beq = dev2(lambda ndx: qs[ndx].b_eq, me)
        o = self.off
        self.Hv, self.Aev, self.Aiv = src[o[0]:o[1]], src[o[1]:o[2]], src[o[2]:o[3]]      # views of the source tensor
        self.dp, self.ds = src[o[3]:o[4]], src[o[4]:o[5]]
        self.Hv.copy_(dev2(lambda ndx: qs[ndx].H.data, self.nnzH))
        self.Aev.copy_(dev2(lambda ndx: qs[ndx].A_eq.data, self.nnzAe))
        self.Aiv.copy_(dev2(lambda ndx: qs[ndx].A_ineq.data, self.nnzAi))
        q0 = qs[0]
        ti = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(dev)      # noqa: E731
        self.Hr, self.Hc = ti(q0.H.row), ti(q0.H.col)
        offd = np.flatnonzero(q0.H.row != q0.H.col)
        self.Hoff = ti(offd)
        self.Aer, self.Aec = ti(q0.A_eq.row), ti(q0.A_eq.col)
        self.Air, self.Aic = ti(q0.A_ineq.row), ti(q0.A_ineq.col)
        self.fsd = ti(self.fs)
        # inertia correction: +coef on the primal rows, -coef on the constraint rows (equality, inequality, link)
        cls = np.zeros(self.nb, dtype=np.int8)
        cls[:n] = 1
        cls[n + mi:] = 2
        solver.set_regularization_classes({ndx: cls for ndx in range(N)})
        self.rhs = solver.new_device_vector()
        self._cnt_bounds = float(sum(np.isfinite(st[k]).sum() for k in ('plb', 'pub', 'ilb', 'iub')))
        self._cnt_duals = float(N * (me + nfs) + N * mi) + self._cnt_bounds
        self.update_kkt_sources()

    # ------------------------------------------------------------------ sizes the loop asks for
    def n_eq_constraints(self):
        return self.N * (self.me + self.nfs)

    def n_ineq_constraints(self):
        return self.N * self.mi

    def set_barrier_parameter(self, barrier):
        self.barrier = float(barrier)

    # ------------------------------------------------------------------ products with the scenario data (shared pattern)
    def _add_rows(self, out, rows, vals):
        return out.index_add_(0, rows, vals)

    def _grad_obj(self):
        import torch
        g = self.c.clone()
        if self.nnzH:
            self._add_rows(g, self.Hr, self.Hv * self.x[self.Hc])
            if self.Hoff.numel():
                o = self.Hoff
                self._add_rows(g, self.Hc[o], self.Hv[o] * self.x[self.Hr[o]])
        return g

    def _jac_products(self):
        """A_eq x - b_eq, A_ineq x, and J^T y of one scenario block (equality, link and inequality parts)."""
        import torch
        eq = -self.beq.clone()
        if self.nnzAe:
            self._add_rows(eq, self.Aer, self.Aev * self.x[self.Aec])
        ineq = torch.zeros_like(self.s)
        if self.nnzAi:
            self._add_rows(ineq, self.Air, self.Aiv * self.x[self.Aic])
        jt = torch.zeros_like(self.x)
        if self.nnzAe:
            self._add_rows(jt, self.Aec, self.Aev * self.yeq[self.Aer])
        if self.nnzAi:
            self._add_rows(jt, self.Aic, self.Aiv * self.yin[self.Air])
        self._add_rows(jt, self.fsd, self.ylink)
        return eq, ineq, jt

    # ------------------------------------------------------------------ the per-iteration producer
    def update_kkt_sources(self):
        """interface.py:450-465: the barrier diagonals, written where the factorisation kernels read them."""
        import torch
        torch.add(self.zl / (self.x - self.lb), self.zu / (self.ub - self.x), out=self.dp)
        torch.add(self.sl / (self.s - self.ilb), self.su / (self.iub - self.s), out=self.ds)

    def evaluate_primal_dual_kkt_matrix(self, timer=None):
        self.update_kkt_sources()
        return self.dk

    def evaluate_primal_dual_kkt_rhs(self, timer=None):
        """interface.py:496-538 + sc_ip_interface.py:1683-1696, into the DeviceBlockVector the solver reads in place."""
        mu = self.barrier
        eq, ineq, jt = self._jac_products()
        n, mi, me, nfs = self.n, self.mi, self.me, self.nfs
        r = self.rhs.group_tensors[self.gid]
        grad_lag_primals = self._grad_obj() + jt - mu / (self.x - self.lb) + mu / (self.ub - self.x)
        r[0:n] = -grad_lag_primals
        r[n:n + mi] = -(-self.yin - mu / (self.s - self.ilb) + mu / (self.iub - self.s))
        r[n + mi:n + mi + me] = -eq
        r[n + mi + me:n + 2 * mi + me] = -(ineq - self.s)
        r[n + 2 * mi + me:] = self.z[:, None] - self.x[self.fsd]
        self.rhs.coupling.copy_(self.ylink[:, :self.B].sum(dim=1))
        return self.rhs

    def regularize_equality_gradient(self, kkt, coef, copy_kkt=True):
        """interface.py:590-609 / sc_ip_interface.py:1736-1745 as a shift of the resident matrix (coef is negative)."""
        w, c, q = kkt.diagonal_shift or (0.0, 0.0, 0.0)
        return kkt.with_diagonal_shift(w, -coef, q)

    def regularize_hessian(self, kkt, coef, copy_kkt=True):
        """interface.py:611-619 / sc_ip_interface.py:1747-1757 (the coupling block gets coef * I as well)."""
        w, c, q = kkt.diagonal_shift or (0.0, 0.0, 0.0)
        # (the reference ADDS coef to the Hessian block of the matrix it was given -- repeated retries accumulate --
        # and REPLACES the coupling block; mirrored so that the iterates are those of the host path)
        return kkt.with_diagonal_shift(w + coef, c, coef)

    # ------------------------------------------------------------------ the step after the solve
    def set_primal_dual_kkt_solution(self, delta):
        """Views of the solution blocks and the bound-dual steps (interface.py:540-588)."""
        d = delta.group_tensors[self.gid]
        d[:, self.B:] = 0.0                                  # padded lanes never limit a step
        n, mi, me = self.n, self.mi, self.me
        self.dx, self.dsl_ = d[0:n], d[n:n + mi]
        self.dyeq, self.dyin = d[n + mi:n + mi + me], d[n + mi + me:n + 2 * mi + me]
        self.dylink = d[n + 2 * mi + me:]
        self.dz = delta.coupling
        mu = self.barrier
        self.dzl = (mu - self.zl * self.dx) / (self.x - self.lb) - self.zl
        self.dzu = (mu + self.zu * self.dx) / (self.ub - self.x) - self.zu
        self.dsl = (mu - self.sl * self.dsl_) / (self.s - self.ilb) - self.sl
        self.dsu = (mu + self.su * self.dsl_) / (self.iub - self.s) - self.su
        for t in (self.dzl, self.dzu, self.dsl, self.dsu):
            t[:, self.B:] = 0.0

    def fraction_to_the_boundary(self, tau):
        """interior_point.py:677-758 with the solver's fused kernel (one pass per variable family)."""
        from parapint_amd.linalg.device_vector_ops import step_stats
        ap1, ad1, _, _ = step_stats(self.solver, self.x, self.dx, self.lb, self.ub, self.zl, self.dzl, self.zu, self.dzu, tau)
        ap2 = ad2 = 1.0
        if self.mi:
            ap2, ad2, _, _ = step_stats(self.solver, self.s, self.dsl_, self.ilb, self.iub, self.sl, self.dsl, self.su,
                                        self.dsu, tau)
        return min(ap1, ap2), min(ad1, ad2)

    def take_step(self, alpha_primal, alpha_dual):
        """interior_point.py:619-626 (solver kernels for the block vectors, the coupling variables with torch)."""
        from parapint_amd.linalg.device_vector_ops import axpy_
        for y, a, x in ((self.x, alpha_primal, self.dx), (self.s, alpha_primal, self.dsl_),
                        (self.yeq, alpha_dual, self.dyeq), (self.yin, alpha_dual, self.dyin),
                        (self.ylink, alpha_dual, self.dylink), (self.zl, alpha_dual, self.dzl),
                        (self.zu, alpha_dual, self.dzu), (self.sl, alpha_dual, self.dsl), (self.su, alpha_dual, self.dsu)):
            if y.numel():
                axpy_(self.solver, y, a, x.contiguous())
        self.z.add_(self.dz, alpha=alpha_primal)

    def check_convergence(self, barrier, error_scaling):
        """interior_point.py:174-317: (primal infeasibility, scaled dual infeasibility, scaled complementarity) with ONE
        transfer of seven scalars."""
        import torch
        B = self.B
        eq, ineq, jt = self._jac_products()
        link = self.x[self.fsd] - self.z[:, None]
        glp = self._grad_obj() + jt - self.zl + self.zu
        glc = -self.ylink[:, :B].sum(dim=1)                  # coupling block of the gradient of the Lagrangian
        gls = -self.yin - self.sl + self.su

        def mx(t):
            return t[:, :B].abs().max() if t.numel() else torch.zeros((), dtype=torch.float64, device=self.x.device)

        def compl(x, bound, dual, lower):
            fin = torch.isfinite(bound)
            mod = torch.where(fin, bound, torch.zeros_like(bound))
            r = ((x - mod) if lower else (mod - x)) * dual - barrier
            return mx(torch.where(fin, r, torch.zeros_like(r)))
        vals = torch.stack([
            torch.max(torch.max(mx(eq), mx(link)), mx(ineq - self.s)),
            torch.max(torch.max(mx(glp), glc.abs().max()), mx(gls)),
            torch.max(torch.max(compl(self.x, self.lb, self.zl, True), compl(self.x, self.ub, self.zu, False)),
                      torch.max(compl(self.s, self.ilb, self.sl, True), compl(self.s, self.iub, self.su, False))),
            self.zl[:, :B].abs().sum() + self.zu[:, :B].abs().sum() + self.sl[:, :B].abs().sum() + self.su[:, :B].abs().sum(),
            self.yeq[:, :B].abs().sum() + self.ylink[:, :B].abs().sum() + self.yin[:, :B].abs().sum()]).cpu().numpy()
        primal_inf, dual_inf, cmpl, bound_sum, dual_sum = (float(v) for v in vals)
        dual_sum += bound_sum
        dual_scaling = max(error_scaling, dual_sum / self._cnt_duals) / error_scaling
        compl_scaling = max(error_scaling, bound_sum / self._cnt_bounds) / error_scaling if self._cnt_bounds > 0 else 1.0
        return primal_inf, dual_inf / dual_scaling, cmpl / compl_scaling

    def evaluate_objective(self):
        g = self._grad_obj()
        c0 = sum(q.c0 for q in self.scenarios)
        return c0 + float((0.5 * ((g + self.c) * self.x)[:, :self.B]).sum().cpu())      # 1/2 x'Hx + c'x = 1/2 (Hx + 2c)'x

    # ------------------------------------------------------------------ results
    def first_stage_solution(self):
        return self.z.cpu().numpy()

    def scenario_primals(self, ndx):
        return self.x[:, self.order.index(ndx)].cpu().numpy()
