"""Device-resident producer of the stochastic KKT system.

The reference's interior-point interfaces rebuild the block KKT matrix and the right-hand side on the host in every
iteration (parapint/interfaces/interface.py:432-538, schur_complement/sc_ip_interface.py:1677-1710,
mpi_sc_ip_interface.py:470-478) and hand them to the linear solver as SciPy / PyNumero objects; ``ip_solve`` then does its
vector work on PyNumero block vectors (algorithms/interior_point.py:174-317, 553-626, 655-758).  For a two-stage stochastic
QP only the barrier diagonals ``z_l / (x - l) + z_u / (u - x)`` (interface.py:450-465) change between iterations; Hessian and
Jacobian values are data.  This class keeps the iterates of this rank's scenarios in HBM in the solver's own [row][instance]
layout -- one state per pattern group of the solver -- and runs everything around the linear solve as HIP kernels
(``csrc/ipstep.hip`` through ``parapint_amd.linalg.device_ip_ops``):

  * ``evaluate_primal_dual_kkt_matrix``: nothing to do -- the step kernel has already written the barrier diagonals where
    the factorisation kernels read them (the value map from KKT entry to source is found once, by evaluating the host
    interface on tagged values);
  * ``evaluate_primal_dual_kkt_rhs``: one elementwise kernel finishes the right-hand side (interface.py:496-538,
    sc_ip_interface.py:1683-1696) in the ``DeviceBlockVector`` the sweeps read in place;
  * ``fraction_to_the_boundary`` / ``take_step`` / ``check_convergence``: bound-dual steps (interface.py:562-588), step
    lengths, the step and the three convergence measures in four kernels + two small reductions; seven scalars reach the
    host through a pinned mailbox;
  * inertia correction: a diagonal shift of the resident matrix (interface.py:590-619, sc_ip_interface.py:1736-1757 ->
    ``DeviceBlockMatrix.with_diagonal_shift``).

Scenarios are ``QuadraticProgram`` objects; they may have several sparsity patterns (one state per pattern group) and are
dealt round-robin over the ranks of the communicator as the reference deals them (mpi_sc_ip_interface.py:14-29).  Between
ranks the step lengths and the measures + coupling block of the right-hand side are all-gathered and combined in rank
order on every rank.  The host class ``StochasticSchurComplementInteriorPointInterface`` stays the specification: the value
map is read off it, and ``tests/test_device_ip.py`` compares the iterates of both.
"""
import zlib

import numpy as np

from parapint_amd.algorithms import interior_point as host_ip
from parapint_amd.interfaces.interface import QuadraticProgram, _relaxed
from parapint_amd.interfaces.schur_complement.sc_ip_interface import StochasticSchurComplementInteriorPointInterface
from parapint_amd.linalg.comm import SerialComm
from parapint_amd.linalg.device_ip_ops import V_HEAD
from parapint_amd.sparse.block_containers import BlockMatrix, MPIBlockMatrix
from parapint_amd.sparse.device_containers import DeviceBlockMatrix

_TAG = 1 << 22          # tags of the five source families are k * _TAG + index + 1 (exact in a double)


class _PatternGroup(object):
    """Scenarios of this rank with one sparsity pattern (Hessian, both Jacobians, first-stage indices)."""

    def __init__(self, q0, fs, ff=(), mapped=False):
        """mapped: the instances tie different coupling variables (time blocks).  fs: variables of the link whose multipliers sit in the block (nonanticipativity; for a time block the start
        states); ff: variables of the forward link of a time block (end states; multipliers in the coupling block)."""
        self.q0, self.fs, self.ff = q0, np.asarray(fs, dtype=np.int64), np.asarray(ff, dtype=np.int64)
        self.members, self.mapped = [], bool(mapped)
        self.nonlinear = False              # the model's own evaluation supplies grad f and c(x): no Hessian terms in the row programs
        self.n, self.me, self.mi, self.nfs, self.nfw = q0.n, q0.A_eq.shape[0], q0.A_ineq.shape[0], self.fs.size, self.ff.size
        self.nb = self.n + 2 * self.mi + self.me + self.nfs          # dimension of one diagonal block
        self.nnzH, self.nnzAe, self.nnzAi = q0.H.nnz, q0.A_eq.nnz, q0.A_ineq.nnz
        # source rows: [H | A_eq | A_ineq | primal barrier diagonal | slack barrier diagonal]
        self.off = np.cumsum([0, self.nnzH, self.nnzAe, self.nnzAi, self.n, self.mi])
        self.nsrc = int(self.off[-1])
        self.value_map = None
        self.K0 = self.A0 = self.A0t = None

    def same_pattern(self, q, fs, ff=()):
        q0 = self.q0

        def same(a, b):
            return a.shape == b.shape and (a.row is b.row or np.array_equal(a.row, b.row)) and \
                (a.col is b.col or np.array_equal(a.col, b.col))
        return same(q.H, q0.H) and same(q.A_eq, q0.A_eq) and same(q.A_ineq, q0.A_ineq) and \
            np.array_equal(np.asarray(fs, dtype=np.int64), self.fs) and np.array_equal(np.asarray(ff, dtype=np.int64), self.ff)

    # ---- value map (once, on the host)
    def find_value_map(self):
        """(src, coef) for the COO entries of K_i followed by those of A_i, read off the HOST interface evaluated on
        tagged values: every Hessian / Jacobian / barrier-diagonal value is replaced by a number that names it -- no
        second description of the KKT layout to keep in step with interfaces/interface.py."""
        q0 = self.q0

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
        vals = np.concatenate([kkt.get_block(0, 0).tocoo().data, self.border_values(kkt.get_block(1, 0).tocoo())])
        a = np.abs(vals)
        fam = (a // _TAG).astype(np.int64)
        idx = (a % _TAG).astype(np.int64) - 1
        is_src = (fam >= 1) & (fam <= 5) & (a == np.round(a)) & (idx >= 0)
        src = np.where(is_src, self.off[np.clip(fam - 1, 0, 4)] + idx, -1).astype(np.int32)
        coef = np.where(is_src, np.sign(vals), vals)
        self.value_map = (src, coef)

    def border_values(self, stochastic_border):
        """Values of the border block A_i in the order of its COO entries (constants).  Mapped groups: the forward link
        (+1 on the end states) and then -1 on the multipliers of the backward link, as DeviceDynamicQPInterface._border
        lists them."""
        if self.mapped:
            return np.concatenate([np.ones(self.nfw), -np.ones(self.nfs)])
        return stochastic_border.data

    # ---- row programs of the kernels (include/parapint_hip.h: pp_ip_group.prog / terms)
    def row_programs(self):
        """(prog [n + me + mi + nfs][4], terms [nt][2]): for the rows grad_x L, A_eq x - b, A_ineq x - s, x_fs - z the
        products {source row or -1, row of W} in the order they are summed -- the Hessian terms of a row first."""
        q, n, mi, me, nfs, nfw, off = self.q0, self.n, self.mi, self.me, self.nfs, self.nfw, self.off
        H, Ae, Ai = q.H, q.A_eq, q.A_ineq
        x0, yeq0, yin0, yl0 = 0, n + mi, n + mi + me, n + 2 * mi + me      # rows of W
        yf0 = self.nb + 2 * n + 2 * mi              # copies of the forward-link multipliers (behind the bound duals)
        offd = np.flatnonzero(H.row != H.col)
        eH, eAe, eAi, k = np.arange(self.nnzH), np.arange(self.nnzAe), np.arange(self.nnzAi), np.arange(nfs)
        kf = np.arange(nfw)
        parts = [   # (program row, source row, row of W, class: 0 Hessian term, 1 other)
            (H.row, off[0] + eH, x0 + H.col, 0),
            (H.col[offd], off[0] + offd, x0 + H.row[offd], 0),              # the mirrored upper triangle
            (Ae.col, off[1] + eAe, yeq0 + Ae.row, 1),                        # A_eq^T y_eq
            (Ai.col, off[2] + eAi, yin0 + Ai.row, 1),                        # A_ineq^T y_ineq
            (self.fs, -np.ones(nfs), yl0 + k, 1),                            # L^T y_link
            (self.ff, -np.ones(nfw), yf0 + kf, 1),                           # L_forward^T rho (time blocks)
            (n + Ae.row, off[1] + eAe, x0 + Ae.col, 1),                      # A_eq x
            (n + me + Ai.row, off[2] + eAi, x0 + Ai.col, 1),                 # A_ineq x
            (n + me + mi + k, -np.ones(nfs), x0 + self.fs, 1),               # x_fs (start states of a time block)
            (n + me + mi + nfs + kf, -np.ones(nfw), x0 + self.ff, 1)]        # end states of a time block
        if self.nonlinear:
            # the model's own evaluation supplies grad f (instead of c + H x) and -c_eq(x) (instead of A_eq x - b): the
            # gradient rows keep J^T y and the link duals, the constraint rows no terms at all, the link rows theirs
            parts = [parts[2], parts[3], parts[4], parts[5], parts[8], parts[9]]
        prow = np.concatenate([np.asarray(p[0], dtype=np.int64) for p in parts])
        src = np.concatenate([np.asarray(p[1], dtype=np.int64) for p in parts])
        wrow = np.concatenate([np.asarray(p[2], dtype=np.int64) for p in parts])
        cls = np.concatenate([np.full(len(p[0]), p[3], dtype=np.int64) for p in parts])
        order = np.lexsort((np.arange(prow.size), cls, prow))              # by row, Hessian terms first, then as listed
        prow, src, wrow, cls = prow[order], src[order], wrow[order], cls[order]
        nprog = n + me + mi + nfs + nfw
        t0 = np.searchsorted(prow, np.arange(nprog), side='left')
        t1 = np.searchsorted(prow, np.arange(nprog), side='right')
        nH = np.bincount(prow[cls == 0], minlength=nprog)[:nprog]
        # execution order: rows that read the same source entries next to each other (a Jacobian entry is read by its
        # constraint row and by the gradient row of its variable: reverse Cuthill-McKee on that graph)
        from scipy.sparse import coo_matrix
        from scipy.sparse.csgraph import reverse_cuthill_mckee
        shared = src >= 0
        by_src = np.argsort(src[shared], kind='stable')
        ps, ss = prow[shared][by_src], src[shared][by_src]
        pair = np.flatnonzero(ss[1:] == ss[:-1])
        if pair.size:
            a, b = ps[pair], ps[pair + 1]
            G = coo_matrix((np.ones(2 * a.size), (np.concatenate([a, b]), np.concatenate([b, a]))), shape=(nprog, nprog)).tocsr()
            order = np.asarray(reverse_cuthill_mckee(G, symmetric_mode=True), dtype=np.int64)
        else:
            order = np.arange(nprog)
        prog = np.stack([t0, t0 + nH, t1, order], axis=1).astype(np.int32)
        terms = np.stack([src, wrow], axis=1).astype(np.int32)
        return prog, terms

    # ---- processed initial points (interior_point.py:433-447, 761-799), all scenarios of a group at once
    def initial_points(self, qs, relax):
        """Rows = scenarios.  The same elementwise operations as the host loop applies per scenario; the slacks
        s = A_ineq x0 (interface.py: init_slacks at the given point) are accumulated entry by entry in the order of the
        COO arrays, as scipy's product does."""
        from scipy.sparse import csr_matrix
        stack = lambda name: np.stack([np.asarray(getattr(q, name), dtype=np.double) for q in qs]) \
            if qs else np.zeros((0, 0))
        lb, ub = _relaxed(stack('lb'), relax, -1.0), _relaxed(stack('ub'), relax, +1.0)
        ilb, iub = _relaxed(stack('ineq_lb'), relax, -1.0), _relaxed(stack('ineq_ub'), relax, +1.0)
        x = stack('x0').copy()
        Ai = self.q0.A_ineq
        prod = np.stack([q.A_ineq.data for q in qs]) * x[:, Ai.col]                 # [scenario][entry]
        by_row = np.argsort(Ai.row, kind='stable')
        gather = csr_matrix((np.ones(self.nnzAi), by_row, np.searchsorted(Ai.row[by_row], np.arange(self.mi + 1))),
                            shape=(self.mi, self.nnzAi))
        s = np.ascontiguousarray((gather @ prod.T).T) if self.mi else np.zeros((len(qs), 0))
        zl, zu = np.ones_like(x), np.ones_like(x)                   # interface.py:263-283 (no ipopt suffixes)
        zl[np.isneginf(stack('lb'))] = 0
        zu[np.isinf(stack('ub'))] = 0
        sl, su = np.zeros_like(s), np.zeros_like(s)
        host_ip.process_init(x, lb, ub)
        host_ip.process_init(s, ilb, iub)
        host_ip.process_init_duals_lb(zl, lb)
        host_ip.process_init_duals_ub(zu, ub)
        host_ip.process_init_duals_lb(sl, ilb)
        host_ip.process_init_duals_ub(su, iub)
        return dict(x=x, s=s, zl=zl, zu=zu, sl=sl, su=su, lb=lb, ub=ub, ilb=ilb, iub=iub)

    def initial_point(self, q, relax):
        return {k: v[0] for k, v in self.initial_points([q], relax).items()}


class _GroupState(object):
    """Device state of one pattern group of the SOLVER (its blocks in lane order)."""
    pass


class DeviceStochasticQPInterface(object):
    """Parameters
    ----------
    scenarios: sequence of QuadraticProgram (only this rank's scenarios are touched; the others may be None)
    first_stage_indices: per scenario the indices of its copies of the first-stage variables, in the order of the
        coupling variables (sc_ip_interface.py:1046-1058)
    comm: communicator of parapint_amd.linalg.comm (None: serial); scenario ndx belongs to rank ndx % size
    """

    def __init__(self, scenarios, first_stage_indices, comm=None, bounds_relaxation_factor=1e-8):
        self.comm = SerialComm() if comm is None else comm
        self.scenarios = list(scenarios)
        self.first_stage_indices = first_stage_indices
        self.N = N = len(self.scenarios)
        if self.comm.size > N:
            raise ValueError('Cannot yet handle more processes than scenarios')     # mpi_sc_ip_interface.py:322-323
        self.local = [ndx for ndx in range(N) if ndx % self.comm.size == self.comm.rank]
        self._relax = bounds_relaxation_factor
        self.nfs = len(first_stage_indices[self.local[0]])
        for ndx in self.local:
            if len(first_stage_indices[ndx]) != self.nfs:
                raise ValueError('every scenario must carry a copy of every first-stage variable')
        self._find_pattern_groups({ndx: (first_stage_indices[ndx], None) for ndx in self.local})

    def _find_pattern_groups(self, links):
        """Pattern groups of the local scenarios / time blocks (index arrays shared between them are recognised by
        identity); links[ndx] = (variables of the link inside the block, variables of the forward link or None)."""
        self.pattern_groups, self.group_of = [], {}
        by_ids, by_sig = {}, {}
        none = np.zeros(0, dtype=np.int64)
        for ndx in self.local:
            q, (fs, ff) = self.scenarios[ndx], links[ndx]
            ids = (q.n, id(q.H.row), id(q.H.col), id(q.A_eq.row), id(q.A_eq.col), id(q.A_ineq.row), id(q.A_ineq.col), id(fs),
                   id(ff))
            pg = by_ids.get(ids)
            if pg is None:
                fsa = np.asarray(fs, dtype=np.int64)
                ffa = none if ff is None else np.asarray(ff, dtype=np.int64)
                sig = (q.n, q.A_eq.shape[0], q.A_ineq.shape[0]) + tuple(
                    (a.size, zlib.crc32(np.ascontiguousarray(a))) for a in (q.H.row, q.H.col, q.A_eq.row, q.A_eq.col,
                                                                           q.A_ineq.row, q.A_ineq.col, fsa, ffa))
                pg = next((g for g in by_sig.get(sig, ()) if g.same_pattern(q, fsa, ffa)), None)
                if pg is None:
                    pg = _PatternGroup(q, fsa, ffa, mapped=ff is not None)
                    pg.find_value_map()
                    by_sig.setdefault(sig, []).append(pg)
                    self.pattern_groups.append(pg)
                by_ids[ids] = pg                 # (`links` keeps the index arrays alive: their ids stay theirs)
            pg.members.append(ndx)
            self.group_of[ndx] = pg
        self.nsrc = max(pg.nsrc for pg in self.pattern_groups)
        self._n_eq = self._n_ineq = None
        self.barrier = None
        self.solver = None
        self.measures = None
        self.states = []
        self._host = None

    # ------------------------------------------------------------------ the host specification (tests, value map)
    @property
    def host(self):
        """The host interface over the same scenarios (built on demand: tests compare with it)."""
        if self._host is None:
            fs = [np.asarray(f, dtype=np.int64) for f in self.first_stage_indices]
            self._host = StochasticSchurComplementInteriorPointInterface(self.scenarios, fs,
                                                                         comm=None if self.comm.size == 1 else self.comm)
            self._host.set_bounds_relaxation_factor(self._relax)
        return self._host

    # ------------------------------------------------------------------ matrix for the symbolic phase
    def device_kkt_matrix(self):
        """DeviceBlockMatrix for do_symbolic_factorization: per pattern group the host KKT blocks of ONE scenario at its
        processed initial point are the pattern (their values fix the static pivot order) -- the solver reads patterns and
        one representative value set from this matrix, the values of all scenarios come from the sources on the device --,
        the value map names the source of every entry."""
        N = self.N
        if self.comm.size > 1:
            owner = -np.ones((N + 1, N + 1), dtype=np.int64)
            for ndx in range(N):
                owner[ndx, ndx] = owner[N, ndx] = owner[ndx, N] = ndx % self.comm.size
            pattern = MPIBlockMatrix(N + 1, N + 1, owner, self.comm)
        else:
            pattern = BlockMatrix(N + 1, N + 1)
        for pg in self.pattern_groups:
            q = pg.q0
            ti = StochasticSchurComplementInteriorPointInterface([q], [pg.fs])
            ti.set_bounds_relaxation_factor(self._relax)
            st = pg.initial_point(q, self._relax)
            nlp = ti.scenario_interface(0)
            nlp.set_primals(st['x']); nlp.set_slacks(st['s'])
            nlp.set_duals_primals_lb(st['zl']); nlp.set_duals_primals_ub(st['zu'])
            nlp.set_duals_slacks_lb(st['sl']); nlp.set_duals_slacks_ub(st['su'])
            kkt = ti.evaluate_primal_dual_kkt_matrix()
            pg.K0, pg.A0 = kkt.get_block(0, 0).tocoo(), kkt.get_block(1, 0).tocoo()
            pg.A0t = pg.A0.transpose().tocoo()
            for ndx in pg.members:
                A = self._border(pg, ndx)
                pattern.set_block(ndx, ndx, pg.K0)
                pattern.set_block(N, ndx, A)
                pattern.set_block(ndx, N, pg.A0t if A is pg.A0 else A.transpose().tocoo())
        for ndx in self.local:
            pattern.set_row_size(ndx, self.group_of[ndx].nb)
            pattern.set_col_size(ndx, self.group_of[ndx].nb)
        corner, sparse_corner, coupling_classes = self._corner()
        pattern.set_block(N, N, corner)
        maps = {ndx: self.group_of[ndx].value_map for ndx in self.local}
        dk = DeviceBlockMatrix(pattern, maps, self.nsrc)
        if sparse_corner:
            dk.Q = corner
        dk.coupling_classes = coupling_classes
        return dk

    def _border(self, pg, ndx):
        """Border block (coupling rows x rows of the diagonal block) of scenario ndx: the same for a whole pattern group."""
        return pg.A0

    def _corner(self):
        """(corner block with explicit zero diagonal, keep it sparse for the solver?, regularisation classes of the
        coupling rows or None: all of them primal variables)."""
        from scipy.sparse import identity
        corner = identity(self.nfs, format='coo')
        corner.data.fill(0)
        return corner, False, None

    # ------------------------------------------------------------------ state on the device
    def attach(self, solver, dk):
        """After do_symbolic_factorization(matrix=dk): iterates, data and bounds of every pattern group of the solver as
        [row][instance] tensors in its lane order, constant KKT values into the source tensors, row programs,
        regularisation classes, right-hand side; then the measures of the initial point."""
        self.solver, self.dk = solver, dk
        self.ops = ops = solver._eng.ip_ops()
        P = self.comm.size
        self.rhs = solver.new_device_vector()
        self.states, descs, classes = [], [], {}
        counts = np.zeros(5)       # finite bounds, duals, sum of c0, equality rows, inequality rows (this rank)
        for gid in sorted(dk.slots):
            order = list(dk.slots[gid])
            pg = self.group_of[order[0]]
            if any(self.group_of[ndx] is not pg for ndx in order):
                raise ValueError('blocks of one solver group come from different pattern groups')
            src = dk.sources[gid]
            B, bpad = len(order), int(src.shape[1])
            n, mi, me, nb, off, nfs, nfw = pg.n, pg.mi, pg.me, pg.nb, pg.off, pg.nfs, pg.nfw
            qs = [self.scenarios[ndx] for ndx in order]
            # One host array [lane][x0 | s0 | lb | ub | ineq_lb | ineq_ub | c | b_eq | H, A_eq, A_ineq values], one copy
            # to the device and one transpose there; the relaxation of the bounds and the processing of the initial
            # point (interior_point.py:433-447, 761-799) run on the device over all lanes at once.
            nvb, nbd, nv = n + mi, 2 * n + 2 * mi, int(off[3])
            raw = np.empty((bpad, nvb + nbd + n + me + nv))
            for b, q in enumerate(qs):
                raw[b] = np.concatenate((q.x0, np.asarray(q.A_ineq @ q.x0, dtype=np.double).ravel(),   # init_slacks
                                         q.lb, q.ub, q.ineq_lb, q.ineq_ub, q.c, q.b_eq, q.H.data, q.A_eq.data, q.A_ineq.data))
            rb = raw[:B, nvb:nvb + nbd]
            counts[0] += np.isfinite(rb).sum()
            counts[2] += sum(q.c0 for q in qs)
            if bpad > B:                                           # padded lanes repeat a real scenario ...
                raw[B:] = raw[0]
                raw[B:, nvb:nvb + n], raw[B:, nvb + n:nvb + 2 * n] = -np.inf, np.inf     # ... but carry no bound (and so
                raw[B:, nvb + 2 * n:nvb + 2 * n + mi], raw[B:, nvb + 2 * n + mi:nvb + nbd] = -np.inf, np.inf   # no bound dual)
            R = ops.rows_from_instances(raw)
            W = ops.zeros((nb + 2 * n + 2 * mi + nfw, bpad))     # (+ copies of the forward-link multipliers, time blocks)
            W[0:nvb] = R[0:nvb]
            bounds = ops.zeros((max(nbd, 1), bpad))
            bounds[0:nbd] = R[nvb:nvb + nbd]
            ops.relax_bounds(bounds, n, mi, self._relax)
            ops.process_initial_point(W, bounds, n, mi, nb)
            data = ops.zeros((max(n + me, 1), bpad))
            data[0:n + me] = R[nvb + nbd:nvb + nbd + n + me]
            if nv > 0:
                src[0:nv] = R[nvb + nbd + n + me:]
            del R
            counts[1] += B * (me + nfs + nfw + mi)
            counts[3] += B * (me + nfs + nfw)
            counts[4] += B * mi
            gs = _GroupState()
            gs.gid, gs.pg, gs.order, gs.B, gs.bpad = gid, pg, order, B, bpad
            gs.W, gs.bounds, gs.data, gs.src = W, bounds, data, src
            gs.G = ops.zeros((max(n, 1), bpad))
            prog, terms = pg.row_programs()
            gs.prog, gs.terms = ops.from_host(prog), ops.from_host(terms if terms.size else np.zeros((1, 2), dtype=np.int32))
            gs.rhs = self.rhs.group_tensors[gid]
            descs.append(dict(n=n, mi=mi, me=me, nfs=nfs, batch=B, bpad=bpad, src_dp=int(off[3]), src_ds=int(off[4]),
                              W=gs.W, bounds=gs.bounds, data=gs.data, src=gs.src, G=gs.G, rhs=gs.rhs, prog=gs.prog,
                              terms=gs.terms))
            self._describe_links(descs[-1], gs)
            # inertia correction: +coef on the primal rows, -coef on the constraint rows (equality, inequality, link)
            cls = np.zeros(nb, dtype=np.int8)
            cls[:n] = 1
            cls[n + mi:] = 2
            for ndx in order:
                classes[ndx] = cls
            self.states.append(gs)
        solver.set_regularization_classes(classes)
        if hasattr(solver, 'warm_device_results'):
            solver.warm_device_results()
        self._prepared = ops.prepare(descs)
        self._ncoup, self._dual_from = self._coupling_layout()
        self.z = ops.zeros((max(self._ncoup - self._dual_from, 1),))         # coupling variables: free, start at 0
        nv = V_HEAD + self._ncoup
        self._alpha_local, self._v_local = ops.zeros((2,)), ops.zeros((nv,))
        self._alpha_table = self._alpha_local if P == 1 else ops.zeros((P, 2))
        self._v_table = self._v_local if P == 1 else ops.zeros((P, nv))
        if P > 1:
            counts = np.asarray(self.comm.allreduce_sum(counts), dtype=np.double)
        self._cnt_bounds, self._cnt_duals, self._c0 = float(counts[0]), float(counts[0] + counts[1]), float(counts[2])
        self._n_eq, self._n_ineq = int(round(counts[3])), int(round(counts[4]))
        self._have_step = False

    def _coupling_layout(self):
        """(entries of the coupling block, first entry that is a coupling VARIABLE): all nfs of them here."""
        return self.nfs, 0

    def _describe_links(self, desc, gs):
        """Which coupling variables the link rows of a group's instances tie: every instance all of them, in order."""
        pass

    # ------------------------------------------------------------------ sizes the loop asks for
    def n_eq_constraints(self):
        return self._n_eq

    def n_ineq_constraints(self):
        return self._n_ineq

    def set_barrier_parameter(self, barrier):
        self.barrier = float(barrier)

    # ------------------------------------------------------------------ the per-iteration producer
    def evaluate_primal_dual_kkt_matrix(self, timer=None):
        """interface.py:432-494: the barrier diagonals are where the factorisation kernels read them already (written by
        the step kernel at the current iterate)."""
        return self.dk

    def evaluate_primal_dual_kkt_rhs(self, timer=None):
        """interface.py:496-538 + sc_ip_interface.py:1683-1696: the variable rows from grad f + J^T y (left by the
        residual kernel) and the barrier parameter; the constraint rows and the coupling block are in place already."""
        self.ops.rhs(self._prepared, self.barrier)
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
        """interface.py:540-588: the step is read where the backward sweep left it; the bound-dual steps are formed on the
        fly by the kernels that use them."""
        for gi, gs in enumerate(self.states):
            self.ops.set_delta(self._prepared, gi, delta.group_tensors[gs.gid])
        self._dz = delta.coupling
        self._have_step = True

    def fraction_to_the_boundary(self, tau):
        """interior_point.py:677-758: this rank's step lengths, then all ranks' (they stay on the device; the host gets
        them with the measures)."""
        self.ops.step_lengths(self._prepared, tau, self.barrier, self._alpha_local)
        self.ops.allgather(self.comm, self._alpha_local, self._alpha_table)

    def take_step(self, unified=False):
        """interior_point.py:619-626 with the step lengths of fraction_to_the_boundary; without a step (the initial
        point) only the barrier diagonals and the elementwise measures are formed."""
        table = self._alpha_table if self._have_step else None
        self.ops.take_step(self._prepared, table, self.comm.size, unified, self.barrier, self.z,
                           self._dz if self._have_step else None)

    def check_convergence(self, error_scaling):
        """interior_point.py:174-317 at the current iterate, for barrier = 0 and for the current barrier parameter in one
        pass: ((primal inf, dual inf, complementarity inf at 0), (..., at mu)), scaled as the reference scales them."""
        ops, P = self.ops, self.comm.size
        ops.residuals(self._prepared, self.z, self._v_local)
        ops.allgather(self.comm, self._v_local, self._v_table)
        ops.publish(self._v_table, self._alpha_table if self._have_step else None, P, self._ncoup, self._dual_from,
                    self.rhs.coupling)
        m = ops.wait()
        bound_sum, dual_sum = float(m[4]), float(m[4] + m[5])
        dual_scaling = max(error_scaling, dual_sum / self._cnt_duals) / error_scaling
        compl_scaling = max(error_scaling, bound_sum / self._cnt_bounds) / error_scaling if self._cnt_bounds > 0 else 1.0
        self.measures = dict(primal_inf=float(m[0]), dual_inf=float(m[1]) / dual_scaling, compl_inf=float(m[2]) / compl_scaling,
                             compl_inf_barrier=float(m[3]) / compl_scaling, objective=float(m[6]) + self._c0,
                             alpha_primal=float(m[7]), alpha_dual=float(m[8]))
        return self.measures

    def evaluate_objective(self):
        return self.measures['objective']

    # ------------------------------------------------------------------ results
    def first_stage_solution(self):
        return self.ops.to_host(self.z)[:self._ncoup - self._dual_from].copy()

    def scenario_primals(self, ndx):
        for gs in self.states:
            if ndx in gs.order:
                return self.ops.to_host(gs.W[:gs.pg.n, gs.order.index(ndx)]).copy()
        raise KeyError('scenario %d is not on this rank' % ndx)


class DeviceDynamicQPInterface(DeviceStochasticQPInterface):
    """The same producer for time-staged problems: device-resident counterpart of
    ``(MPI)DynamicSchurComplementInteriorPointInterface`` (sc_ip_interface.py:13-1026, mpi_sc_ip_interface.py:32-270) for
    time blocks given as QuadraticPrograms.  A time block ties its start states to the coupling states before it (backward
    link, multipliers inside the block) and its end states to the coupling states after it (forward link, multipliers in the
    coupling block), so the instances of a pattern group tie DIFFERENT coupling variables: the kernels get per-instance
    offsets (include/parapint_hip.h: mapped groups), every instance carries a copy of the multipliers of its forward link,
    and the coupling block [rho | z] of the right-hand side is scattered instead of summed over the lanes.

    Parameters
    ----------
    time_blocks: sequence over the time blocks of (QuadraticProgram, start states, end states) -- what
        ``build_model_for_time_block`` returns (sc_ip_interface.py:107-143); entries of other ranks may be None
    comm: communicator of parapint_amd.linalg.comm (None: serial); time block ndx belongs to rank ndx % size
    """

    def __init__(self, time_blocks, comm=None, bounds_relaxation_factor=1e-8):
        self.comm = SerialComm() if comm is None else comm
        self.time_blocks = list(time_blocks)
        self.N = T = len(self.time_blocks)
        if self.comm.size > T:
            raise ValueError('Cannot yet handle more processes than time blocks')    # mpi_sc_ip_interface.py:79-80
        if T < 2:
            raise ValueError('a time-staged problem needs at least two time blocks (one has no coupling states)')
        self.local = [ndx for ndx in range(T) if ndx % self.comm.size == self.comm.rank]
        self._relax = bounds_relaxation_factor
        self.scenarios = [None if blk is None else blk[0] for blk in self.time_blocks]
        self.num_states = ns = len(self.time_blocks[self.local[0]][1])
        self.ncz = ns * (T - 1)
        self.nfs = 2 * self.ncz                       # rows of the coupling block: forward multipliers, coupling states
        none = np.zeros(0, dtype=np.int64)
        links = {}
        for ndx in self.local:
            _, start, end = self.time_blocks[ndx]
            if len(start) != ns or len(end) != ns:
                raise ValueError('every time block must name the same number of start and end states')
            links[ndx] = (start if ndx != 0 else none, end if ndx != T - 1 else none)
        self._find_pattern_groups(links)

    @property
    def host(self):
        """The host interface over the same time blocks (tests compare with it)."""
        if self._host is None:
            from parapint_amd.interfaces.schur_complement.sc_ip_interface import DynamicSchurComplementInteriorPointInterface
            blocks = self.time_blocks

            class _Given(DynamicSchurComplementInteriorPointInterface):
                def build_model_for_time_block(self, ndx, start_t, end_t, add_init_conditions):
                    return blocks[ndx]
            self._host = _Given(0.0, 1.0, self.N, comm=None if self.comm.size == 1 else self.comm)
            self._host.set_bounds_relaxation_factor(self._relax)
        return self._host

    def _border(self, pg, ndx):
        """(2 ncz x dim K_ndx): +1 of the forward link on the rows rho_ndx, -1 on the multipliers of the backward link on
        the rows z_{ndx-1} (sc_ip_interface.py:308-327); the entry order is that of _PatternGroup.border_values."""
        from scipy.sparse import coo_matrix
        ns, ncz = self.num_states, self.ncz
        rows = np.concatenate([ns * ndx + np.arange(pg.nfw), ncz + ns * (ndx - 1) + np.arange(pg.nfs)])
        cols = np.concatenate([pg.ff, pg.n + 2 * pg.mi + pg.me + np.arange(pg.nfs)])
        return coo_matrix((pg.border_values(None), (rows.astype(np.int64), cols.astype(np.int64))), shape=(2 * ncz, pg.nb))

    def _corner(self):
        """[[0, -I], [-I, 0]] (:329-357), sparse: 2 n_s (T - 1) rows; the rho rows are constraint rows (class 2)."""
        from scipy.sparse import coo_matrix
        m = self.ncz
        i = np.arange(m)
        corner = coo_matrix((-np.ones(2 * m), (np.concatenate([m + i, i]), np.concatenate([i, m + i]))), shape=(2 * m, 2 * m))
        return corner, True, np.concatenate([np.full(m, 2, dtype=np.int8), np.full(m, 1, dtype=np.int8)])

    def _coupling_layout(self):
        return 2 * self.ncz, self.ncz

    def _describe_links(self, desc, gs):
        ns, T = self.num_states, self.N
        lanes = list(gs.order) + [gs.order[0]] * (gs.bpad - gs.B)          # (padded lanes: any valid offsets)
        zoff = np.zeros((2, gs.bpad), dtype=np.int32)
        zoff[0] = [ns * (t - 1) if t > 0 else 0 for t in lanes]
        zoff[1] = [ns * t if t < T - 1 else 0 for t in lanes]
        gs.zoff = self.ops.from_host(zoff)
        desc.update(nfw=gs.pg.nfw, ncz=self.ncz, zoff=gs.zoff)

    def coupling_states(self):
        """The states between the time blocks (n_states * (T - 1)), block after block."""
        return self.first_stage_solution()


class _DeviceModelMixin(object):
    """What turns one of the QP producers into the producer of a NONLINEAR problem class: the constant QP data are replaced
    by what the caller's DEVICE MODEL writes after every step (the functions Pyomo / ASL evaluate for the reference at every
    iterate, interfaces/interface.py:432-538, here on the device so that the iterate never leaves it).

    The subproblems are objects with the NLP protocol of interfaces/interface.py (equality constraints and bounds); they
    give sizes, bounds, the initial point and the sparsity patterns (Hessian of the Lagrangian: lower triangle in the order
    of ``evaluate_hessian_lag()``, Jacobian in the order of ``evaluate_jacobian_eq()``).
    device_model: callable (NLP objects of a pattern group in lane order, padded batch) -> object with
        ``evaluate(W, src, data, layout)`` that, from the primals W[0:n] and the equality multipliers
        W[layout['y_eq']:...] of every lane, writes the Hessian values into src[layout['hess']:...], the Jacobian values
        into src[layout['jac']:...], grad f into data[0:n], -c_eq(x) into data[n:n+me] and the objective value of every
        lane into data[layout['obj_row']] -- all [row][lane] arrays; called after every step, on the solver's stream."""

    @staticmethod
    def _qp_standin(nlp):
        """The QP that shares the model's patterns and its values at the initial point (pattern, pivot order, first values)."""
        from scipy.sparse import coo_matrix
        if nlp.n_ineq_constraints() != 0:
            raise NotImplementedError('device-resident nonlinear models: equality constraints and bounds only')
        x0 = np.asarray(nlp.init_primals(), dtype=np.double)
        nlp.set_primals(x0)
        nlp.set_duals_eq(np.asarray(nlp.init_duals_eq(), dtype=np.double))
        H, A = coo_matrix(nlp.evaluate_hessian_lag()), coo_matrix(nlp.evaluate_jacobian_eq())
        if np.any(H.row < H.col):
            raise ValueError('evaluate_hessian_lag() of a device-resident model must return the lower triangle')
        qp = QuadraticProgram(c=np.asarray(nlp.evaluate_grad_objective(), dtype=np.double), A_eq=A,
                              b_eq=A @ x0 - np.asarray(nlp.evaluate_eq_constraints(), dtype=np.double),
                              lb=nlp.primals_lb(), ub=nlp.primals_ub(), H=None, x0=x0)
        qp.H = H                                             # (entry order kept: the device model writes values in it)
        return qp

    def _mark_nonlinear(self):
        for pg in self.pattern_groups:
            pg.nonlinear = True

    def _describe_links(self, desc, gs):
        super(_DeviceModelMixin, self)._describe_links(desc, gs)
        pg, ops = gs.pg, self.ops
        # one more data row: the objective value of every lane, from the model's own evaluation
        data = ops.zeros((pg.n + pg.me + 1, gs.bpad))
        data[0:pg.n + pg.me] = gs.data[0:pg.n + pg.me]
        gs.data = data
        desc.update(data=data, obj_row=pg.n + pg.me)
        lanes = list(gs.order) + [gs.order[0]] * (gs.bpad - gs.B)
        gs.model = self._device_model([self._nlps[t] for t in lanes], gs.bpad)
        gs.layout = dict(n=pg.n, me=pg.me, y_eq=pg.n + pg.mi, hess=int(pg.off[0]), jac=int(pg.off[1]), obj_row=pg.n + pg.me,
                         batch=gs.B, bpad=gs.bpad)

    def attach(self, solver, dk):
        super(_DeviceModelMixin, self).attach(solver, dk)
        self._c0 = 0.0                                           # (the objective comes whole from the model)

    def take_step(self, unified=False):
        super(_DeviceModelMixin, self).take_step(unified)
        for gs in self.states:                                   # the model at the new iterate, before the residual kernels
            gs.model.evaluate(gs.W, gs.src, gs.data, gs.layout)


class DeviceStochasticNLPInterface(_DeviceModelMixin, DeviceStochasticQPInterface):
    """Two-stage stochastic programs with NONLINEAR scenario problems and device-resident iterates (see _DeviceModelMixin).
    scenarios: NLP objects (entries of other ranks may be None); first_stage_indices as for DeviceStochasticQPInterface."""

    def __init__(self, scenarios, first_stage_indices, device_model, comm=None, bounds_relaxation_factor=1e-8):
        self._device_model = device_model
        self._nlps = {ndx: nlp for ndx, nlp in enumerate(scenarios) if nlp is not None}
        qps = [None if nlp is None else self._qp_standin(nlp) for nlp in scenarios]
        DeviceStochasticQPInterface.__init__(self, qps, first_stage_indices, comm=comm,
                                             bounds_relaxation_factor=bounds_relaxation_factor)
        self._mark_nonlinear()


class DeviceDynamicNLPInterface(_DeviceModelMixin, DeviceDynamicQPInterface):
    """Time-staged NONLINEAR problems with device-resident iterates (see _DeviceModelMixin): time_blocks per time block
    (nlp, start states, end states) -- what ``build_model_for_time_block`` returns; entries of other ranks may be None."""

    def __init__(self, time_blocks, device_model, comm=None, bounds_relaxation_factor=1e-8):
        self._device_model = device_model
        self._nlps, blocks = {}, []
        for ndx, blk in enumerate(time_blocks):
            if blk is None:
                blocks.append(None)
                continue
            nlp, start, end = blk
            self._nlps[ndx] = nlp
            blocks.append((self._qp_standin(nlp), start, end))
        DeviceDynamicQPInterface.__init__(self, blocks, comm=comm, bounds_relaxation_factor=bounds_relaxation_factor)
        self._mark_nonlinear()
