"""MI355X-native drop-in for parapint's Schur-complement linear solvers.

``HipSchurComplementLinearSolver`` replaces
``parapint.linalg.MPISchurComplementLinearSolver``
(parapint/linalg/schur_complement/mpi_explicit_schur_complement.py:128-452) and, given a
plain ``BlockMatrix``, its serial twin ``SchurComplementLinearSolver``
(explicit_schur_complement.py:16-177).  Same constructor arguments, same five
``LinearSolverInterface`` methods with the same keyword names, same status / exception
behaviour, same ownership rule (Q10), so ``parapint.algorithms.ip_solve`` and the
Schur-complement interior-point interfaces call it unchanged.

What differs underneath:
  * the per-block sub-solvers (MA27 / MUMPS / SciPy wrappers) are not called: all local blocks
    are factorised together on the GPU by hand-written HIP kernels reached through the C ABI of
    include/parapint_hip.h (``subproblem_solvers`` / ``schur_complement_solver`` are accepted
    and kept for API compatibility only);
  * S is formed by partial factorisation of [[K_i, A_i^T], [A_i, 0]] instead of n_c
    single-right-hand-side solves per block, and is a dense device buffer;
  * the mpi4py collectives are one RCCL all-reduce of that buffer (+ packed status/inertia) per
    numeric factorisation and one of the n_c coupling right-hand side per back-solve,
    issued through the injected communicator (parapint_amd.linalg.comm).
Lower triangle of every K_i is authoritative (MA27 semantics, quirk Q5).
There is no CPU fallback: without the HIP library / a GPU the constructor raises.
"""
import numpy as np

from parapint_amd.linalg.base_linear_solver_interface import LinearSolverInterface
from parapint_amd.linalg.comm import SerialComm, default_comm
from parapint_amd.linalg.results import LinearSolverResults, LinearSolverStatus

_OK = (LinearSolverStatus.successful, LinearSolverStatus.warning)

from parapint_amd.linalg._solver_support import (HipEngine, _BY_SEVERITY, _S8, _SEVERITY, _BlockInfo, _Group, _Labels,  # noqa: F401
                                                  _NullTimer, _PatternChanged, _UnionMatrix, _addr, _canonical, _coo, _flat,
                                                  _index_intact, _index_record, _roctx)
from parapint_amd.linalg.coupling_structure import CouplingStructureMixin
from parapint_amd.sparse.block_containers import BlockMatrix as _BlockMatrix, MPIBlockMatrix as _MPIBlockMatrix

from parapint_amd.linalg.host_staging import HostStagingMixin, _F8, _OWN_MATRICES      # noqa: F401
from parapint_amd.linalg.pivot_repair import PivotRepairMixin
from parapint_amd.linalg import general_blocks
from parapint_amd.linalg.solution_check import SolutionCheckMixin


class HipSchurComplementLinearSolver(PivotRepairMixin, SolutionCheckMixin, CouplingStructureMixin, HostStagingMixin,
                                    LinearSolverInterface):
    """Solve A x = b for block-bordered-diagonal symmetric A (lower border supplied)::

          K1          transpose(A1)
              K2      transpose(A2)
                  K3  transpose(A3)
          A1  A2  A3  Q

    Parameters
    ----------
    subproblem_solvers, schur_complement_solver:
        accepted for signature compatibility with the reference (mpi_...:154-155); unused.
    comm:
        communicator (parapint_amd.linalg.comm); default: torch.distributed if initialised, else serial.
    engine:
        numeric engine; default ``HipEngine`` (GPU).  Tests inject a host interpreter to rehearse the
        multi-rank host logic on CPU -- the product never does.
    """

    @classmethod
    def getLoggerName(cls):
        return 'hip_schur_complement'

    def __init__(self, subproblem_solvers=None, schur_complement_solver=None, comm=None, engine=None,
                 memory_budget_bytes=None, result_buffers=0, pivot_tolerance=None, symbolic_pivot_threshold=None,
                 general_blocks=None):
        self.subproblem_solvers = subproblem_solvers
        self.schur_complement_solver = schur_complement_solver
        # General-LU semantics (general_blocks.py): the sub-solver objects the reference's callers hand over are never
        # called here, but a ScipyInterface among them says what the caller expects of unsymmetric blocks -- SuperLU reads
        # both triangles (scipy_interface.py:26-31).  On that route every host matrix is looked at (symmetry up to rounding of the
        # local diagonal blocks and of the corner, O(nnz) on the host, agreed across the ranks) and an unsymmetric one is
        # factorised through its symmetric embedding; symmetric matrices and every other route take the path below.
        if general_blocks is None:
            subs = list(subproblem_solvers.values()) if isinstance(subproblem_solvers, dict) else list(subproblem_solvers or [])
            general_blocks = any(getattr(o, 'general_lu', False) for o in subs + [schur_complement_solver])
        self._general_route = bool(general_blocks)
        self._general_mode = None           # True: the last symbolic / numeric phase went through the embedding
        self._general_inner = None
        self.comm = default_comm() if comm is None else comm
        self._eng = HipEngine() if engine is None else engine
        # cap on the device value storage (factor panels, work vectors); None = no cap.  A plan that needs more makes
        # the numeric phase return not_enough_memory until increase_memory_allocation() has raised the cap -- the
        # reallocation protocol of interior_point.py:634-652 / ma27_interface.py:126-131, 153-154
        if memory_budget_bytes is not None:
            self._eng.set_memory_budget(memory_budget_bytes)
        # MA27's cntl(1) (ma27_interface.py:36-47; examples/stochastic.py:120-124).  `pivot_tolerance` u, if given, is
        # enforced on every instance of every factorisation as |l_ij| <= 1/u: an instance beyond it makes the
        # factorisation refresh the static pivot order from that instance's values and, failing that, report `singular`
        # for the inertia-correction loop to regularise.  Not enforced by default: interior-point matrices grow by 1/mu
        # benignly (a slack pivot of 1e-9 against its -1 coupling).  `symbolic_pivot_threshold` is the u of the static
        # 1x1 / 2x2 choice on the representative values (default 0.01).
        if pivot_tolerance is not None or symbolic_pivot_threshold is not None:
            self._eng.set_pivot_tolerance(0.0 if symbolic_pivot_threshold is None else symbolic_pivot_threshold,
                                          0.0 if pivot_tolerance is None else pivot_tolerance)
        self._growth_guard = bool(pivot_tolerance)
        # Opt-in (`refresh_thresholds`, e.g. (0.1, 0.3)): a static pivot sequence that has broken down is chosen again
        # with a STRICTER 1x1 test (more 2x2 pivots) at the first and at every later refresh, as MA27 users raise cntl(1)
        # when a factorisation turns out fragile (Ipopt: ma27_pivtol -> ma27_pivtolmax).  MEASURED on the QP of
        # bench.py's ip_loop: with 1024 scenarios 155 iterations / 183 refreshes at 0.01 throughout against 66 / 6 with
        # (0.1, 0.3) -- but with 256 scenarios 45 / 16 against, depending on rounding noise in the iterates, 46-48 / 6 or
        # a run that ends in 'Exceeded maximum inertia correction' (2x2 pivots over two primal variables can cancel for
        # other barrier values; 1x1 pivots on positive diagonals cannot).  Off by default.
        self._u_user = (0.0 if symbolic_pivot_threshold is None else float(symbolic_pivot_threshold),
                        0.0 if pivot_tolerance is None else float(pivot_tolerance))
        self.refresh_thresholds = ()
        self._u_symbolic_now = self._u_user[0]
        self._mapped = False                # blocks have local coupling rows + maps (dynamic problems)
        self._btd = None                    # (block size, blocks) of a block-tridiagonal S, else None: dense
        self._btd_sequential = False        # eliminate its blocks in ascending order instead of by cyclic reduction
        self._cperm = self._cinv = None     # ordering of the coupling variables under which S is block tridiagonal
        self._dense_coupling_limit = 1024   # a mapped S up to this dimension stays dense
        self._classes = None                # regularisation classes by block index (kept across re-plans)
        self._constant_entries = None       # declare_constant_entries: {block index: (constK, constA)} (kept across re-plans)
        self._constant_check = None
        self._device_maps = None            # (nsrc, value maps by block index) of a DeviceBlockMatrix (f2)
        self._dev_results = []
        self._dev_turn = 0
        # result_buffers = 0 (default): do_back_solve returns fresh arrays / tensors, as the reference does -- results of
        # different calls never alias.  result_buffers = k > 0 is an opt-in for callers that consume a result before the
        # k-th following back-solve (bench.py, the interior-point loops): views of k page-locked host buffers (D2H at
        # PCIe speed, no 75 MB allocation per call at the headline size) / k device vectors used in turn
        self._result_buffers = max(0, int(result_buffers))
        self.block_dim = 0
        self.block_matrix = None
        self.local_block_indices = []
        self._groups = []
        self._binfo = {}
        self._nc = 0
        self._inertia = None
        self._num_status = None
        self._pattern_only = False
        self._have_classes = False
        self.pivot_order_refreshes = 0      # numeric factorisations that needed a new static pivot sequence
        self.pivot_order_refreshes_since_symbolic = 0
        self.refreshes_skipped = 0          # breakdowns reported as `singular` without a new sequence (futile lately)
        self.refresh_causes = {'zero_pivot': 0, 'growth': 0, 'residual': 0}      # ... because of a zero pivot / of element growth in a block
        self.diagonal_shift_refactorizations = 0      # factorisations from resident values + a diagonal shift (f1)
        self._last_Q = None
        self._base_Q = None
        self._last_error = ''
        self.growth_instances = 0           # instances of the last factorisation with a factor entry beyond 1 / u (1e8 if no guard)
        self.plan_stats = []
        # A pivot-order refresh that does not cure a breakdown is repeated at every later breakdown by default: a
        # static sequence that breaks does not mean the matrix is singular, and the reference's sub-solvers (dynamic
        # pivoting) never report a regular matrix singular.  refresh_backoff = True (opt-in; problems whose Jacobian is
        # rank deficient at EVERY iterate) reports the next 2^k - 1 breakdowns (at most 63) of a group as `singular` at
        # once after the k-th futile refresh in a row -- no symbolic phase (0.2 s at C3) per factorisation.  A
        # breakdown on exactly the values of the last futile refresh is never planned again either way.
        self.refresh_backoff = False
        self._last_device_base = None       # DeviceBlockMatrix of the last full factorisation (diagonal-shift fast path)
        self._refreshed = []                # groups whose pivot order the last refresh chose again
        # Instances of ONE pattern group that need incompatible static pivot sequences (MA27 pivots every block on its own
        # values, ma27_interface.py:110-140): when the sequence chosen from instance A breaks on instance B and the one
        # chosen from B breaks on A, the group is split -- the conflicting blocks move to a VARIANT of the pattern group
        # (same pattern, a plan of its own from its own first block).  Host containers only: the lane layout of device
        # containers belongs to their producer.  max_pivot_repairs bounds the extra factorisations of one call.
        self.split_conflicting_groups = True
        self.max_pivot_repairs = 6
        self.max_group_variants = 3
        self.group_splits = 0               # blocks moved to a variant group so far
        self._variant = {}                  # block index -> variant of its pattern group (0: none)
        self._maps_checked = None           # value maps already compared block by block
        self._layout_cache = None           # (binfo, layout, dims) of device_layout()
        self._cinv_t = self._rc_pad = self._xc_pad = None       # device copies of the coupling order (block-tridiagonal S)
        self._cperm_pad = None
        self._prefetch_rhs = None           # DeviceBlockVector whose forward sweep rides behind the factorisations (prefetch_forward)
        self._forward_done_for = None       # ... and the one whose forward sweep the last factorisation has already enqueued
        self._prefetch_before_factor = False   # the announcement preceded the block phase of the running factorisation
        # Index arrays recognised by identity (the fast paths of the host boundary) are checked for in-place rewrites:
        # size and address at every call; a CRC of their contents at every call for the first `pattern_check_bytes` of
        # distinct arrays (blocks that share their index arrays -- one Jacobian structure, one object -- are always
        # covered: 0.3 MB at the headline size) and at every `pattern_check_interval`-th numeric factorisation for all
        # of them (1: every call; 0: never beyond the byte budget).
        self.pattern_check_interval = 8
        self.pattern_check_bytes = 4 << 20
        self.constant_sample_blocks = 2     # declare_constant_entries(check=None): blocks per group and call whose declared entries are compared (0.1 ms each at C3)
        self._stage_calls = 0
        self._index_records = {}            # id(index array) -> (array, size, address, checksum)
        self._index_sets = {}               # ids of a block's four index arrays -> the tuple of them (shared by blocks)
        self._deferred_solve = None         # (do_back_solve_deferred: the back-solve whose verdict confirm_solution() still has to collect)
        self.solution_changed_on_confirm = False
        self._init_solution_check()         # every back-solve is checked on the device, refined, repaired (solution_check.py)

    # ------------------------------------------------------------------ helpers
    def _local_blocks(self, matrix):
        nb = self.block_dim
        own = getattr(matrix, 'rank_ownership', None)
        if own is None:
            return list(range(nb - 1))
        rank = self.comm.rank
        return [ndx for ndx in range(nb - 1)
                if own[ndx, ndx] == rank or (own[ndx, ndx] == -1 and rank == 0)]   # mpi_...:199-203

    def _agree_status(self, status):
        """Rank-consistent status (mpi_...:19-30): the most severe status of any rank wins."""
        if self.comm.size == 1:
            return status
        v = int(self.comm.allreduce_max(np.array([_SEVERITY[status]], dtype=np.int64))[0])
        return _BY_SEVERITY[v]

    def _guarded(self, res, fn, *args):
        """Runs an engine step unless an earlier one failed; an exception that carries a C status becomes that status
        (so `raise_on_error=False` callers -- the reallocation retry and the inertia loop -- get a status, and a rank
        that failed still reaches the collectives)."""
        if res.status not in _OK:
            return None
        try:
            return fn(*args)
        except Exception as err:
            status = getattr(err, 'status', None)
            if status is None:
                raise
            res.status = LinearSolverStatus(status)
            self._last_error = str(err)
            return None

    def _build_groups(self, matrix):
        last = self.block_dim - 1
        nc = matrix.get_row_size(last) if hasattr(matrix, 'get_row_size') else matrix.get_block(last, last).shape[0]
        self._nc = int(nc)
        groups, by_sig, by_ids, same_by_ids, binfo = [], {}, {}, {}, {}
        all_zero = True
        empty_i, empty_d = np.zeros(0, dtype=np.int32), np.zeros(0)
        # first pass: which coupling rows does every block touch?  If every block touches all of them (the union of the
        # cliques of a stochastic program is dense, mpi_...:88-125) the groups are uniform; otherwise (time blocks of a
        # dynamic problem, scenarios with a subset of the first-stage variables) every block gets LOCAL coupling rows and a
        # map to the global ones, so that blocks with the same local structure still share one plan and one batch
        fetched = {}
        uniform = True
        seen = {}          # (id(K block), id(A block)) -> what was read off them: callers may hand one object to many blocks
        for ndx in self.local_block_indices:
            Kb, A = matrix.get_block(ndx, ndx), matrix.get_block(last, ndx)
            hit = seen.get((id(Kb), id(A)))
            if hit is not None:
                fetched[ndx] = hit
                uniform = uniform and hit[7].size == self._nc
                continue
            kr, kc, kd, kshape = _coo(Kb)
            if kshape[0] != kshape[1]:
                raise ValueError('Matrix must be square')
            br, bc, bd = (empty_i, empty_i, empty_d) if A is None else _coo(A)[:3]
            used = np.unique(br)
            if used.size != self._nc:
                uniform = False
            # (Kb and A ride along: a temporary built by get_block must stay alive, or its id could be handed out again)
            fetched[ndx] = seen[(id(Kb), id(A))] = (kr, kc, kd, kshape[0], br, bc, bd, used, Kb, A)
        # every rank must take the same layout (a mapped rank joins collectives of its own in _coupling_structure, and a
        # dense and a block-tridiagonal S cannot meet in one all-reduce): one rank's non-uniform blocks decide for all
        if self.comm.size > 1:
            uniform = int(self.comm.allreduce_max(np.array([0 if uniform else 1], dtype=np.int64))[0]) == 0
        self._mapped = (not uniform) and self._nc > 0
        for ndx in self.local_block_indices:
            kr, kc, kd, n, br, bc, bd, used = fetched[ndx][:8]
            m = self._nc
            cmap = None
            if self._mapped:
                cmap = used.astype(np.int32)
                m = cmap.size
                br_global = br
                br = np.searchsorted(cmap, br).astype(np.int32)
            # (index arrays shared by many blocks -- one Jacobian structure, one object -- are hashed once)
            var = self._variant.get(ndx, 0)      # (a block moved out of its pattern group: a group of its own kind)
            ids = None if self._mapped else (n, m, id(kr), id(kc), id(br), id(bc), var)
            g = by_ids.get(ids) if ids is not None else None
            if g is None:
                raw_sig = (n, m, kr.tobytes(), kc.tobytes(), br.tobytes(), bc.tobytes(), var)
                g = by_sig.get(raw_sig)
                if g is not None and ids is not None:
                    by_ids[ids] = g
            if g is None:
                rowK, colK, cpK, ciK = _canonical(kr, kc, n, True)
                rowB, colB, cpB, ciB = _canonical(br, bc, n, False)
                can_sig = (n, m, rowK.tobytes(), colK.tobytes(), rowB.tobytes(), colB.tobytes(), var)
                g = by_sig.get(can_sig)
                if g is None:
                    can_ptr = np.concatenate([cpK, cpB[1:] + cpK[-1]]).astype(np.int32)
                    can_idx = np.concatenate([ciK, ciB + kd.size]).astype(np.int32)
                    g = _Group(n, rowK, colK, rowB, colB, can_ptr, can_idx, kd.size, kd.size + bd.size,
                               (kr.copy(), kc.copy(), br.copy(), bc.copy()))
                    g.gid = len(groups)
                    g.m = m
                    g.cmaps = []
                    groups.append(g)
                    by_sig[can_sig] = g
                by_sig[raw_sig] = g
                if ids is not None:
                    by_ids[ids] = g
            bi = _BlockInfo()
            bi.group, bi.slot, bi.n = g, len(g.blocks), n
            bi.seen = None
            bi.cmap = cmap
            bi.br_cache = None if cmap is None else ((br_global.__array_interface__['data'][0], br_global.size), br, br_global)
            # blocks whose raw COO order differs from the group's reference order are
            # canonicalised on the host at every numeric call (quirk Q7)
            ref = g.raw_refs
            same = same_by_ids.get(ids) if ids is not None else None
            if same is None:
                same = (kr.size == ref[0].size and br.size == ref[2].size and
                        np.array_equal(kr, ref[0]) and np.array_equal(kc, ref[1]) and
                        np.array_equal(br, ref[2]) and np.array_equal(bc, ref[3]))
                if ids is not None:
                    same_by_ids[ids] = same
            bi.raw_sig = same
            g.blocks.append(ndx)
            g.cmaps.append(cmap)
            binfo[ndx] = bi
            if g.rep_vals is None:
                raw = np.concatenate([kd, bd])
                vals = self._canonical_values(g, raw, kr, kc, br, bc, bi.raw_sig)
                if np.any(vals[:g.rowK.size] != 0.0):
                    g.rep_vals = vals
            if np.any(kd != 0.0):
                all_zero = False
        self._coupling_structure(matrix, groups)
        pinned = getattr(self._eng, 'alloc_pinned', None)

        def alloc(shape, pinned_only=False):
            if pinned is not None:
                return pinned(shape)
            return None if pinned_only else np.zeros(shape, dtype=np.double)
        for g in groups:
            g._alloc, g.result_buffers = alloc, self._result_buffers
            g.x_shape = (len(g.blocks), g.n)
            g.x_turn = 0
        self._groups, self._binfo = groups, binfo
        self._index_records = {}
        self._index_sets = {}
        self._pattern_only = any(g.rep_vals is None for g in groups)
        return all_zero

    def _run_symbolic(self):
        self._cinv_t = self._rc_pad = self._xc_pad = None      # (device copies of the coupling order: per plan)
        self._dev_results = []
        if self._btd is not None:
            self.plan_stats = self._eng.symbolic(self._btd[0] * self._btd[1], self._groups, btd=self._btd, cinv=self._cinv)
            if hasattr(self._eng, 'set_coupling_schedule'):
                self._eng.set_coupling_schedule(self._btd_sequential)
        else:
            self.plan_stats = self._eng.symbolic(self._nc, self._groups)
        self._have_classes = False
        if self._device_maps is not None:     # value maps are per plan, too
            self._apply_value_maps()
        if self._classes is not None:         # classes are per plan: apply them to the new one
            try:
                self._apply_classes()
            except Exception:
                # a classed row without a diagonal entry in the new plan: the fast path stays off until
                # set_regularization_classes is called again; the ordinary path is unaffected
                self._have_classes = False
        if self._constant_entries is not None:       # (library groups are made anew by every plan)
            self._apply_constant_entries()

    def _apply_value_maps(self):
        nsrc, maps = self._device_maps
        checked = self._maps_checked is maps       # (a pivot-order refresh: same maps, same groups)
        for g in self._groups:
            src, coef = maps[g.blocks[0]]
            for ndx in (() if checked else g.blocks[1:]):
                s2, c2 = maps[ndx]
                if not ((s2 is src or np.array_equal(src, s2)) and (c2 is coef or np.array_equal(coef, c2))):
                    raise ValueError('blocks of one pattern group must share one value map (block %d differs)' % ndx)
            if len(src) != g.nraw:
                raise ValueError('value map of block %d has %d entries, the block has %d' % (g.blocks[0], len(src), g.nraw))
            self._eng.set_value_map(g.gid, nsrc, src, coef)
        self._maps_checked = maps

    def device_layout(self):
        """{block index: (group id, lane)} of the local blocks, and {group id: (batch, padded batch, block dimension)}."""
        cached = self._layout_cache
        if cached is None or cached[0] is not self._binfo:       # (a new dict per symbolic phase / re-plan)
            layout = {ndx: (bi.group.gid, bi.slot) for ndx, bi in self._binfo.items()}
            dims = {g.gid: (len(g.blocks), -(-len(g.blocks) // 64) * 64, g.n) for g in self._groups}
            cached = self._layout_cache = (self._binfo, layout, dims)
        return cached[1], cached[2]

    def new_device_vector(self, zero=True):
        """A DeviceBlockVector with the structure of this solver's right-hand sides (zero-filled; zero=False: left
        uninitialised, for a result every entry of which the backward sweep writes)."""
        from parapint_amd.sparse.device_containers import DeviceBlockVector
        layout, dims = self.device_layout()
        v = DeviceBlockVector(self.block_dim, layout)
        new = self._eng.new_tensor if zero or not hasattr(self._eng, 'new_tensor_uninitialized') else \
            self._eng.new_tensor_uninitialized
        for gid, (batch, bpad, n) in dims.items():
            v.group_tensors[gid] = new((n, bpad))
        v.coupling = self._eng.new_tensor((max(self._nc, 1),))[:self._nc]
        return v

    def device_vector_from_host(self, bv):
        import torch
        v = self.new_device_vector()
        for g in self._groups:
            host = np.zeros(tuple(v.group_tensors[g.gid].shape))
            for slot, ndx in enumerate(g.blocks):
                host[:, slot] = _flat(bv.get_block(ndx))
            v.group_tensors[g.gid].copy_(torch.from_numpy(host))
        if self._nc > 0:
            v.coupling.copy_(torch.from_numpy(np.ascontiguousarray(_flat(bv.get_block(self.block_dim - 1)))))
        return v

    def _apply_classes(self):
        for g in self._groups:
            self._eng.set_diagonal_classes(g.gid, self._classes[g.blocks[0]])
        self._have_classes = True

    def _replan_union(self, matrix):
        """New plan on (planned pattern) U (pattern of `matrix`), values of `matrix`; later matrices with either
        pattern are subsets and need no further planning."""
        from scipy.sparse import coo_matrix
        last = self.block_dim - 1
        blocks = {}
        for ndx in self.local_block_indices:
            g = self._binfo[ndx].group
            n = g.n
            kr, kc, kd, _ = _coo(matrix.get_block(ndx, ndx))
            low = kr >= kc
            rows = np.concatenate([g.rowK, kr[low]])
            cols = np.concatenate([g.colK, kc[low]])
            data = np.concatenate([np.zeros(g.rowK.size), kd[low]])
            blocks[(ndx, ndx)] = coo_matrix((data, (rows, cols)), shape=(n, n))   # duplicates are summed by the plan
            A = matrix.get_block(last, ndx)
            if A is None:
                br, bc, bd = np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32), np.zeros(0)
            else:
                br, bc, bd, _ = _coo(A)
            old_rows = g.rowB if not self._mapped else self._binfo[ndx].cmap[g.rowB]     # back to global coupling rows
            rows = np.concatenate([old_rows, br])
            cols = np.concatenate([g.colB, bc])
            data = np.concatenate([np.zeros(g.rowB.size), bd])
            blocks[(last, ndx)] = coo_matrix((data, (rows, cols)), shape=(self._nc, n))
        blocks[(last, last)] = matrix.get_block(last, last)     # (the structure of S depends on the pattern of Q, too)
        self._build_groups(_UnionMatrix(self.block_dim, blocks, self._nc))
        self._run_symbolic()
        self._pattern_only = False

    # ------------------------------------------------------------------ interface
    def do_symbolic_factorization(self, matrix, raise_on_error=True, timer=None):
        nbrows, nbcols = matrix.bshape
        if nbrows != nbcols:
            raise ValueError('The block matrix provided is not square.')
        self.block_dim = nbrows
        self.local_block_indices = self._local_blocks(matrix)
        self._general_mode = self._unsymmetric_anywhere(matrix)
        if self._general_mode:
            self._inertia = self._num_status = None
            return self._general_solver().do_symbolic_factorization(
                general_blocks.embed_matrix(matrix, self.local_block_indices), raise_on_error=raise_on_error, timer=timer)
        timer = _Labels(timer)
        self._inertia = None
        self._num_status = None
        self._classes = None
        self._constant_entries = None       # (declarations are about the entries of ONE symbolic phase)
        self._dev_results = []
        device_matrix = hasattr(matrix, 'value_maps')
        if self._u_symbolic_now != self._u_user[0] and hasattr(self._eng, 'set_pivot_tolerance'):
            self._eng.set_pivot_tolerance(*self._u_user)        # (a new problem starts from the caller's threshold again)
        self._u_symbolic_now = self._u_user[0]
        self.pivot_order_refreshes_since_symbolic = 0
        self._variant = {}
        self._last_device_base = None
        self._device_maps = (matrix.nsrc, matrix.value_maps) if device_matrix else None
        self._symbolic_pattern = getattr(matrix, 'pattern', matrix)      # (HostValueMatrix: values over this very object)
        self._flat_rows = {}
        self._maps_checked = None           # (new groups: the maps of their blocks are compared again)
        res = LinearSolverResults(LinearSolverStatus.successful)
        timer.start('factorize')
        self._guarded(res, self._build_groups, matrix.pattern if device_matrix else matrix)
        self._guarded(res, self._run_symbolic)
        if device_matrix:
            self._guarded(res, self._attach_device_matrix, matrix)
        timer.stop('factorize')
        res.status = self._agree_status(res.status)
        if res.status not in _OK:
            if raise_on_error:
                raise RuntimeError('Symbolic factorization unsuccessful; status: ' + str(res.status))
            return res
        # the structure of S (mpi_...:228-255): the union of the blocks' border cliques is dense for a stochastic
        # program, so S is a dense n_c x n_c device buffer; the four sub-steps of the reference have no work left
        timer.start('sc_structure')
        for label in ('build_border_matrices', 'gather_all_nonzero_elements', 'construct_schur_complement',
                      'get_sc_data_slices'):
            timer.start(label)
            timer.stop(label)
        timer.stop('sc_structure')
        return res

    def do_numeric_factorization(self, matrix, raise_on_error=True, timer=None):
        if self._general_route and self._general_mode is not None:
            unsym = self._unsymmetric_anywhere(matrix)
            if unsym != self._general_mode:
                # (the values have turned (un)symmetric since the symbolic phase: the other path's symbolic phase, on this
                # matrix -- same pattern by the interface's contract)
                res = self.do_symbolic_factorization(matrix, raise_on_error=raise_on_error)
                if res.status not in _OK:
                    return res
            if unsym:
                res = self._general_inner.do_numeric_factorization(
                    general_blocks.embed_matrix(matrix, self.local_block_indices), raise_on_error=raise_on_error, timer=timer)
                self._num_status = res.status
                self._inertia = None
                return res
        if self._deferred_solve is not None:
            self.confirm_solution()         # (a deferred back-solve is never left unjudged: its factors are still in place here)
        shift = getattr(matrix, 'diagonal_shift', None)
        if shift is not None and self._have_classes and matrix.base is self._last_device_base:
            # "the last matrix + a diagonal": one retry of the inertia-correction loop (interior_point.py:377-392) from
            # the values resident on the device -- reached through the reference's unchanged call site
            return self.refactorize_with_diagonal_shift(shift[0], shift[1], coupling_shift=shift[2],
                                                        raise_on_error=raise_on_error, timer=timer,
                                                        coupling_classes=getattr(matrix, 'coupling_classes', None))
        if shift is not None and shift != (0.0, 0.0, 0.0):
            raise RuntimeError('a shifted device matrix needs set_regularization_classes and a factorisation of its base first')
        if shift is not None:
            matrix = matrix.base
        if hasattr(matrix, 'value_maps'):
            self._last_device_base = matrix
        self._last_factor_call = ('full', matrix)
        res = self._numeric_factorization(matrix, timer)
        if res.status == LinearSolverStatus.singular and self._refresh_pivot_order():
            # a block broke down under the static pivot sequence: it was fixed from the values the symbolic phase
            # saw, and these values differ enough to need another one.  MA27 pivots dynamically and would not report
            # this matrix singular unless it is (ma27_interface.py:124-136), so order again with the values that
            # broke and factorise once more before the inertia-correction loop is told `singular`.
            res = self._numeric_factorization(matrix, timer)
            self._note_refresh_outcome(res.status != LinearSolverStatus.singular)
            repairs = 0
            # (a HostValueMatrix is spelled out as COO blocks for the rebuild of the groups: rare, and the values matter)
            spelled = [None]

            def with_values():
                if spelled[0] is None:
                    spelled[0] = matrix.to_block_matrix(self.local_block_indices) if getattr(matrix, 'flat_values', None) is not None else matrix
                return spelled[0]
            while (res.status == LinearSolverStatus.singular and self.split_conflicting_groups and
                   not hasattr(matrix, 'value_maps') and repairs < self.max_pivot_repairs and
                   self._split_conflicting(with_values())):
                # the new sequence broke on ANOTHER instance of the group: two instances that need different sequences
                repairs += 1
                res = self._numeric_factorization(matrix, timer)
        if res.status not in _OK and raise_on_error:
            raise RuntimeError('Numeric factorization unsuccessful; status: ' + str(res.status))
        return res

    def _numeric_factorization(self, matrix, timer=None):
        timer = _Labels(timer)
        if self.block_dim == 0:
            raise RuntimeError('Perform symbolic factorization first!')
        self.block_matrix = matrix
        res = LinearSolverResults(LinearSolverStatus.successful)
        timer.start('form SC')
        self._prefetch_before_factor = self._prefetch_rhs is not None       # (announced before the block phase is enqueued)
        timer.start('factorize')
        if hasattr(matrix, 'value_maps'):
            self._guarded(res, self._bind_device_matrix, matrix)      # f2: values are gathered from the device sources
        else:
            timer.start('values to device')
            self._guarded(res, self._stage_and_upload, matrix)
            timer.stop('values to device')
        self._guarded(res, self._eng.numeric_factor_blocks)
        timer.stop('factorize')
        # the n_c solves + products per block of the reference (mpi_...:312-333) are the coupling rows of the same
        # partial factorisation; what is left is their outer products
        timer.start('back solve')
        # with a forward sweep announced (prefetch_forward) the Schur update leaves the handle's stream to it -- unless S is
        # all-reduced through torch.distributed, which enqueues on that stream
        side = self._prefetch_rhs is not None and (self.comm.size == 1 or getattr(self._eng, '_direct_rccl', lambda c: False)(self.comm))
        if side:
            self._guarded(res, self._eng.numeric_schur, True)
        else:
            self._guarded(res, self._eng.numeric_schur)
        timer.stop('back solve')
        timer.start('dot product')
        timer.stop('dot product')
        Q = matrix.Q if hasattr(matrix, 'value_maps') else self._guarded(res, self._coupling_block, matrix)
        if Q is not None and self._btd is None and hasattr(Q, 'toarray'):
            Q = Q.toarray()
        self._base_Q = Q
        return self._finish_numeric(res, Q, timer)

    def _attach_device_matrix(self, matrix):
        """Source tensors ([nsrc][padded batch], zero-filled) and the lane order of every group, for the producer."""
        matrix.sources, matrix.slots = {}, {}
        for g in self._groups:
            bpad = -(-len(g.blocks) // 64) * 64
            matrix.sources[g.gid] = self._eng.new_tensor((matrix.nsrc, bpad))
            matrix.slots[g.gid] = list(g.blocks)
        self._bind_device_matrix(matrix)

    def _bind_device_matrix(self, matrix):
        if self._device_maps is None or matrix.value_maps is not self._device_maps[1]:
            raise RuntimeError('this matrix was not given to do_symbolic_factorization (value maps differ)')
        for g in self._groups:
            t = matrix.sources.get(g.gid)
            bpad = -(-len(g.blocks) // 64) * 64
            if t is None or tuple(t.shape) != (matrix.nsrc, bpad) or not t.is_contiguous():
                raise ValueError('source tensor of group %d must be a contiguous [%d][%d] float64 device tensor' %
                                 (g.gid, matrix.nsrc, bpad))
            self._eng.bind_source_tensor(g.gid, t)
            g.device_sources = t

    def _from_coupling_order(self, v):
        return v if self._btd is None else np.ascontiguousarray(v[self._cinv])

    def _coupling_block(self, matrix):
        last = self.block_dim - 1
        Qb = matrix.get_block(last, last)
        Q = None
        if Qb is not None:
            Qc = Qb.tocoo()
            if Qc.nnz > 0 and np.any(Qc.data != 0.0):
                if self._btd is not None:                   # (a block-tridiagonal S can be far too large for a dense Q)
                    import scipy.sparse as _sp
                    low = _sp.tril(Qc).tocsr()
                    return (low + _sp.tril(Qc, -1).T).tocoo()
                Q = Qc.toarray()
                Q = np.tril(Q) + np.tril(Q, -1).T          # lower triangle authoritative
        return Q

    def _finish_numeric(self, res, Q, timer):
        """Second half of a numeric factorisation, shared with the diagonal-shift fast path: status agreement BEFORE the
        collective (a rank whose block phase failed must not leave the others waiting in the all-reduce; the reference
        gathers the sub-solver statuses first, mpi_...:294-305), all-reduce of S, dense factor, status + inertia."""
        if res.status not in _OK:
            if self.comm.size == 1:
                for label in ('communicate',):
                    timer.start(label)
                    timer.stop(label)
                timer.stop('form SC')
                self._inertia = None
                self._num_status = res.status
                return res
            # this rank's block phase failed on the host side: it contributes a zero S whose tail carries the failure,
            # so that the one all-reduce below also agrees the status (no collective of its own, nobody left waiting)
            local = res.status
            res.status = LinearSolverStatus.successful
            self._guarded(res, self._eng.fail_local, local.value if local.value in (1, 2, 3) else 3)
            if res.status not in _OK:
                raise RuntimeError('rank %d cannot take part in the collective after a failure: %s' %
                                   (self.comm.rank, self._last_error))
        timer.start('communicate')
        for label in ('zeros', 'Barrier'):      # S is zeroed on the device; the collective is stream-ordered
            timer.start(label)
            timer.stop(label)
        timer.start('Allreduce')
        self._guarded(res, self._eng.allreduce_schur, self.comm)
        timer.stop('Allreduce')
        timer.start('add')                      # (+ Q happens inside the dense factor kernel)
        timer.stop('add')
        timer.stop('communicate')
        timer.stop('form SC')
        timer.start('factor SC')
        if self._btd is not None:
            self._guarded(res, self._eng.factor_schur_corner, *self._btd_corner(Q))
        else:
            self._guarded(res, self._eng.factor_schur, Q)
        self._last_Q = Q
        self._forward_done_for = None
        if self._prefetch_rhs is not None and res.status in _OK:
            # the forward sweep of the announced right-hand side does not depend on S: enqueued here, before the host waits
            # for the status, it runs beside the one-workgroup factorisation of S (a stream of its own in the library)
            # (announced before the factorisation was enqueued: with several pattern groups and a block-tridiagonal S each
            # group's sweep follows its own block factorisation, beside the Schur update and the cyclic reduction)
            self._guarded(res, self._forward_sweep, self._prefetch_rhs, self._prefetch_before_factor)
            if res.status in _OK:
                self._forward_done_for = self._prefetch_rhs
        elif self._prefetch_rhs is not None and self.comm.size > 1:
            # a host-side failure of THIS rank's S phase (e.g. an allocation of the dense factorisation): the other ranks
            # enqueue the all-reduce of r_s of the announced sweep now -- join it, so that the collectives stay paired
            try:
                self._eng.allreduce_rs(self.comm)
            except Exception:
                pass
        st = self._guarded(res, self._eng.status)
        if (st is not None and st[0] == 2 and self._btd is not None and not self._btd_sequential and
                hasattr(self._eng, 'set_coupling_schedule')):
            # a singular diagonal block in the odd-even order of cyclic reduction does not mean S is singular (S is
            # indefinite): eliminate in ascending block order instead, and keep doing so (same decision on every rank:
            # all of them hold the same all-reduced S)
            self._btd_sequential = True
            self._guarded(res, self._eng.set_coupling_schedule, True)
            self._guarded(res, self._eng.factor_schur_corner, *self._btd_corner(Q))
            st = self._guarded(res, self._eng.status)
        timer.stop('factor SC')
        if st is not None:
            status, pos, neg, zero = st
            if self._btd is not None:
                pos -= self._btd[0] * self._btd[1] - self._nc        # the unit diagonal of the padding rows
            self._inertia = (pos, neg, zero)
            res.status = LinearSolverStatus(status)
            self.growth_instances = self._eng.growth_count()
        else:
            self._inertia = None
        # after the all-reduce every rank holds the same block counts, the same failure tail and the same S: the device
        # status is rank-consistent as it is; only a host-side failure of the dense phase itself can differ
        if self.comm.size > 1 and st is None:
            # (the other ranks read a successful status from their mailboxes and make no further collective in which this
            # failure could be agreed: end loudly instead of returning a status only this rank holds)
            raise RuntimeError('rank %d: the factorisation of S failed on the host side (%s; status %s); the other ranks '
                               'cannot learn of it' % (self.comm.rank, self._last_error, res.status))
        self._num_status = res.status
        return res

    # ------------------------------------------------------------------ inertia-correction fast path
    def set_regularization_classes(self, classes):
        """Which rows of every local K_i the inertia-correction loop shifts: ``classes[ndx]`` is an int8 array over
        the rows of block ndx -- 0 none, 1 Hessian row (+coef, interfaces/interface.py:611-619), 2 constraint row
        (-coef, interface.py:590-609 / sc_ip_interface.py:1736-1757).  Blocks of one pattern group must agree.
        Every classed row needs its diagonal entry in the planned pattern (it is there once a regularised matrix has
        been factorised: the plan is then on the union pattern)."""
        if self._num_status is None and not self._groups:
            raise RuntimeError('Perform symbolic factorization first!')
        keep = {}
        for g in self._groups:
            ref = None
            for ndx in g.blocks:
                c = np.ascontiguousarray(classes[ndx], dtype=np.int8)
                if c.size != g.n:
                    raise ValueError('classes of block %d have length %d, expected %d' % (ndx, c.size, g.n))
                if ref is None:
                    ref = c
                elif not np.array_equal(ref, c):
                    raise ValueError('blocks of one pattern group must have identical regularization classes')
                keep[ndx] = ref
        self._classes = keep
        self._apply_classes()

    def refactorize_with_diagonal_shift(self, delta_w, delta_c, coupling_shift=0.0, raise_on_error=True, timer=None,
                                        _retry=False, coupling_classes=None):
        """Numeric factorisation of (the last matrix given to do_numeric_factorization) + delta_w on the classed
        Hessian diagonals - delta_c on the classed constraint diagonals + coupling_shift * I on the coupling block,
        from the values already resident on the device: what one retry of the inertia-correction loop
        (interior_point.py:377-386) needs, without rebuilding, staging or uploading the KKT matrix."""
        if self._general_mode:
            raise RuntimeError('refactorize_with_diagonal_shift: the last matrix went through the symmetric embedding of an '
                               'unsymmetric matrix (general_blocks.py), which has no regularisation classes')
        timer = _Labels(timer)
        if self._num_status is None:
            raise RuntimeError('Perform numeric factorization first!')
        if not self._have_classes:
            raise RuntimeError('Call set_regularization_classes first!')
        res = LinearSolverResults(LinearSolverStatus.successful)
        self._last_factor_call = ('shift', (delta_w, delta_c, coupling_shift, coupling_classes))
        self.diagonal_shift_refactorizations += 1
        timer.start('form SC')
        self._prefetch_before_factor = self._prefetch_rhs is not None       # (announced before the block phase is enqueued)
        timer.start('factorize')
        self._guarded(res, self._eng.numeric_local_shifted, delta_w, delta_c)
        timer.stop('factorize')
        Q = None if self._base_Q is None else self._base_Q.copy()
        # diagonal of the coupling block: + coupling_shift (rows of class 1; all rows without classes), - delta_c (class 2)
        if coupling_classes is None:
            cdiag = np.full(self._nc, float(coupling_shift))
        else:
            cls = np.asarray(coupling_classes)
            if cls.shape != (self._nc,):
                raise ValueError('coupling_classes must name every coupling row')
            cdiag = np.where(cls == 1, float(coupling_shift), np.where(cls == 2, -float(delta_c), 0.0))
        if np.any(cdiag != 0.0) and self._nc > 0:
            if self._btd is not None:
                import scipy.sparse as _sp
                shift = _sp.diags(cdiag, format='coo')
                Q = shift if Q is None else (_sp.coo_matrix(Q) + shift).tocoo()
            else:
                Q = (np.zeros((self._nc, self._nc)) if Q is None else Q) + np.diag(cdiag)
        base = self._base_Q
        res = self._finish_numeric(res, Q, timer)
        self._base_Q = base                     # shifts are relative to the matrix of the last full factorisation
        if res.status == LinearSolverStatus.singular and not _retry and self._refresh_pivot_order((delta_w, delta_c)):
            # as in do_numeric_factorization: a static pivot sequence that broke down is chosen again from the values
            # (shift included) of the instance that broke, before the caller is told `singular`
            if self._last_device_base is not None and self._device_maps is not None:
                self._bind_device_matrix(self._last_device_base)      # (the new plan has device buffers of its own)
            else:
                # host COO matrices: the values of the last full factorisation are still in the pinned staging arrays;
                # the new plan's device buffers are empty (pp_begin_symbolic frees the groups), so send them again
                for g in self._groups:
                    self._eng.upload_values_compact(g.gid, g.staging)
            res = self.refactorize_with_diagonal_shift(delta_w, delta_c, coupling_shift=coupling_shift,
                                                       raise_on_error=False, timer=timer, _retry=True,
                                                       coupling_classes=coupling_classes)
            self._note_refresh_outcome(res.status != LinearSolverStatus.singular)
            if res.status not in _OK and raise_on_error:
                raise RuntimeError('Numeric factorization unsuccessful; status: ' + str(res.status))
            return res
        if res.status not in _OK and raise_on_error:
            raise RuntimeError('Numeric factorization unsuccessful; status: ' + str(res.status))
        return res

    def do_back_solve(self, rhs, timer=None, _repairs=None):
        if self._general_mode:
            if self._num_status is None:
                raise RuntimeError('Perform numeric factorization first!')
            loc, nb = self.local_block_indices, self.block_dim
            xbar = self._general_inner.do_back_solve(general_blocks.embed_vector(rhs, loc, nb), timer=timer)
            self.last_multiplier_norm = general_blocks.residual_of_multipliers(xbar, loc, nb)
            return general_blocks.extract_solution(xbar, rhs, loc, nb)
        if timer is None:
            timer = _NullTimer()
        if self._num_status is None:
            raise RuntimeError('Perform numeric factorization first!')
        if self._deferred_solve is not None:
            self.confirm_solution()
        if _repairs is None:
            _repairs = self.max_solve_repairs
            self._repairs_this_solve = 0
        if hasattr(rhs, 'group_tensors'):
            return self._device_back_solve(rhs, timer, _repairs)
        timer.start('back_solve')
        last = self.block_dim - 1
        if hasattr(self._eng, 'bind_native_vectors'):
            for g in self._groups:
                self._eng.bind_native_vectors(g.gid, None, None)
        copy_rows = getattr(self._eng, 'copy_rows', None)
        upload_rows = getattr(self._eng, 'upload_rhs_rows', None)
        pending = {}
        get = rhs.get_block
        binfo = self._binfo
        nested = False
        timer.start('rhs to device')
        for ndx in self.local_block_indices:
            bi = binfo[ndx]
            v = get(ndx)
            if copy_rows is not None and type(v) is np.ndarray and v.dtype.char == 'd' and v.strides == _S8 and v.size == bi.n:
                p = pending.get(bi.group.gid)
                if p is None:
                    p = pending[bi.group.gid] = (bi.group, [], [])
                p[1].append(bi.slot)
                p[2].append(v)
            else:
                nested = nested or hasattr(v, 'get_block')
                bi.group.rhs_staging[bi.slot] = _flat(v)
        uploaded = set()
        for g, slots, vecs in pending.values():
            if upload_rows is not None and len(slots) == len(g.blocks) and slots == list(range(len(slots))):
                upload_rows(g, vecs)                        # staged and sent slice by slice, copies overlapping
                uploaded.add(g.gid)
            else:
                copy_rows(g.rhs_staging, list(zip(slots, vecs)))       # 75 MB at the headline size: on the library's host threads
        for g in self._groups:
            if g.gid not in uploaded:
                self._eng.upload_rhs(g.gid, g.rhs_staging)
        timer.stop('rhs to device')
        timer.start('solve')
        self._eng.solve_forward()
        self._eng.allreduce_rs(self.comm)
        rc = self._to_coupling_order(_flat(rhs.get_block(last))) if self._nc > 0 else None
        self._eng.solve_coupling(rc)
        self._eng.solve_backward()
        if self._checking():
            # (solution_check.py: residual of every local block on the device, refinement, a new pivot sequence if need be)
            bad = self._verify_solution(bc_host=rc)
            if bad is not None:
                if _repairs > 0 and self._repair_after_inaccurate_solve(bad):
                    timer.stop('solve')
                    timer.stop('back_solve')
                    return self.do_back_solve(rhs, timer, _repairs - 1)
                self._give_up_on_solution(bad)
        timer.stop('solve')
        timer.start('solution to host')
        xout = {}
        rows_dl = getattr(self._eng, 'download_solution_rows', None)
        in_flight = False
        for g in self._groups:
            # one array per group and call: its rows are handed out as the result blocks (no per-block copy).  Default:
            # a fresh array, so that results of different calls never alias -- filled through a pinned array by host
            # threads; result_buffers = k > 0: k pinned arrays handed out in turn, written by the copy engine itself.
            if g.x_pool:
                xout[g.gid] = g.x_pool[g.x_turn % len(g.x_pool)]
                g.x_turn += 1
                if rows_dl is not None:
                    rows_dl(g, xout[g.gid])                 # asynchronous: the result structure is built meanwhile
                    in_flight = True
                else:
                    self._eng.download_solution(g.gid, xout[g.gid])
            else:
                xout[g.gid] = np.empty(g.x_shape, dtype=np.double)
                if rows_dl is not None and getattr(self._eng, 'alloc_pinned', None) is not None:
                    if g.x_pinned is None:
                        g.x_pinned = self._eng.alloc_pinned(g.x_shape)
                    rows_dl(g, g.x_pinned, xout[g.gid])
                else:
                    self._eng.download_solution(g.gid, xout[g.gid])
        # (mpi_...:390 uses copy_structure(); every local block and the coupling block are set below and non-local
        # blocks stay unset either way, so a container that can skip the zero-filled placeholders is asked to)
        result = rhs.copy_structure_unset() if hasattr(rhs, 'copy_structure_unset') else rhs.copy_structure()
        set_block = result.set_block
        if nested:
            for ndx in self.local_block_indices:
                bi = binfo[ndx]
                x = xout[bi.group.gid][bi.slot]
                blk = get(ndx)
                if hasattr(blk, 'get_block'):          # nested BlockVector (quirk Q9)
                    if in_flight:
                        self._eng.synchronize()
                        in_flight = False
                    out = blk.copy_structure()
                    out.copyfrom(x)
                    x = out
                set_block(ndx, x)
        else:
            for ndx in self.local_block_indices:
                bi = binfo[ndx]
                set_block(ndx, xout[bi.group.gid][bi.slot])
        coupling = self._from_coupling_order(self._eng.coupling_solution())
        if in_flight:
            self._eng.synchronize()
        timer.stop('solution to host')
        blk = rhs.get_block(last)
        if hasattr(blk, 'get_block'):
            out = blk.copy_structure()
            out.copyfrom(coupling)
            coupling = out
        result.set_block(last, coupling)
        timer.stop('back_solve')
        return result

    def warm_device_results(self):
        """Creates the result vectors of result_buffers > 0 and the library's value storage (factor panels, work vectors:
        0.85 GB at C3) now -- set-up time -- instead of in the first factorisation and back-solve."""
        if self._result_buffers > 0 and not self._dev_results:
            self._dev_results = [self.new_device_vector() for _ in range(self._result_buffers)]
        if hasattr(self._eng, 'bind_native_vectors'):
            try:
                for g in self._groups:
                    self._eng.bind_native_vectors(g.gid, None, None)       # (allocates: include/parapint_hip.h, pp_bind_native_vectors)
            except Exception as err:
                if getattr(err, 'status', None) is None:
                    raise              # (over budget: the numeric factorisation reports it, the caller's reallocation loop acts)

    def prefetch_forward(self, rhs):
        """Opt-in for callers that know the right-hand side of the next back-solve before they factorise (an interior-point
        iteration does): every numeric factorisation from now on enqueues the forward sweep of `rhs` (a DeviceBlockVector,
        read in place; it must not change until the back-solve) behind its block phase, where it overlaps the dense
        factorisation of S and the host's wait for the status; ``do_back_solve(rhs)`` with the same object then starts at
        the coupling solve.  ``prefetch_forward(None)`` ends it; so does a back-solve."""
        if rhs is not None and not hasattr(rhs, 'group_tensors'):
            raise ValueError('prefetch_forward takes a DeviceBlockVector')
        self._prefetch_rhs = rhs
        self._prefetch_before_factor = False
        self._forward_done_for = None

    def _forward_sweep(self, rhs, early=False):
        for g in self._groups:
            self._eng.bind_native_vectors(g.gid, rhs.group_tensors[g.gid], None)
        self._eng.solve_forward(early=early)
        self._eng.allreduce_rs(self.comm)

    def do_back_solve_deferred(self, rhs, timer=None):
        """do_back_solve for a DeviceBlockVector WITHOUT the wait for its a-posteriori check: everything, the check included,
        is enqueued and the result vector returned; ``confirm_solution()`` then waits for the verdict and refines / repairs
        in place as do_back_solve would have.  For callers that have further work to enqueue which only READS the solution (an
        interior-point iteration: the step lengths) and must not lose the stream while the host waits.  The solution may be
        handed on or modified only after confirm_solution(); every other entry point of the solver confirms first."""
        if not hasattr(rhs, 'group_tensors'):
            return self.do_back_solve(rhs, timer)
        if self._num_status is None:
            raise RuntimeError('Perform numeric factorization first!')
        self.confirm_solution()
        self._repairs_this_solve = 0
        return self._device_back_solve(rhs, _NullTimer() if timer is None else timer, self.max_solve_repairs, defer=True)

    def confirm_solution(self):
        """Collects the verdict of a deferred back-solve (refinement / repair as in do_back_solve); returns its result vector
        (the same object unless a repair had to solve again into a fresh one), or None if nothing was pending.
        ``solution_changed_on_confirm`` tells whether the vector's contents changed."""
        pend, self._deferred_solve = self._deferred_solve, None
        self.solution_changed_on_confirm = False
        if pend is None:
            return None
        rhs, out, rc_dev, repairs, hand_over = pend
        before = (self.refinement_steps, self.solve_repairs)
        bad = self._verify_solution(bc_dev=rc_dev, begun=True)
        if bad is not None:
            if repairs > 0 and self._repair_after_inaccurate_solve(bad):
                if self._result_buffers > 0:
                    self._dev_turn -= 1            # (the same result vector again)
                out = self._device_back_solve(rhs, _NullTimer(), repairs - 1)
            else:
                self._give_up_on_solution(bad)
        elif self.refinement_steps != before[0]:
            hand_over()
        self.solution_changed_on_confirm = (self.refinement_steps, self.solve_repairs) != before
        return out

    def _device_back_solve(self, rhs, timer, _repairs=0, defer=False):
        """do_back_solve for a DeviceBlockVector: right-hand sides are read where they are, the solution is written
        into a fresh device vector (or, with result_buffers = k > 0, into k vectors handed out in turn); no host copies."""
        timer.start('back_solve')
        if self._result_buffers == 0:
            out = self.new_device_vector(zero=False)        # default: results of different calls never alias
        else:
            if not self._dev_results:
                self._dev_results = [self.new_device_vector() for _ in range(self._result_buffers)]
            out = self._dev_results[self._dev_turn % len(self._dev_results)]
            self._dev_turn += 1
        for g in self._groups:
            self._eng.bind_native_vectors(g.gid, rhs.group_tensors[g.gid], out.group_tensors[g.gid])
        if self._forward_done_for is not rhs:
            self._eng.solve_forward()
            self._eng.allreduce_rs(self.comm)
        self._prefetch_rhs = self._forward_done_for = None
        rc_dev = rhs.coupling if self._nc > 0 else None
        if self._btd is not None and rc_dev is not None:
            # r_s into the (padded) ordering under which S is block tridiagonal, x_s back: two small library kernels
            if self._cinv_t is None or self._cinv_t.numel() != self._nc:
                self._cinv_t = self._eng.index_tensor(self._cinv)
                self._rc_pad = self._eng.new_tensor((self._btd[0] * self._btd[1],))
                self._xc_pad = self._eng.new_tensor((self._btd[0] * self._btd[1],))
            self._eng.permute(self._cinv_t, rc_dev, self._rc_pad, scatter=True)
            rc_dev = self._rc_pad
        self._eng.solve_coupling_dev(rc_dev)
        self._eng.solve_backward()

        def hand_over_coupling():
            if self._nc > 0:
                if self._btd is not None:
                    self._eng.copy_coupling_solution(self._xc_pad)
                    self._eng.permute(self._cinv_t, self._xc_pad, out.coupling, scatter=False)
                else:
                    self._eng.copy_coupling_solution(out.coupling)
        hand_over_coupling()             # (enqueued before the host waits for the verdict below; again after a refinement)
        if defer and self._checking():
            self._check_bc_host, self._check_bc_dev = None, rc_dev
            self._rho_begin()
            self._deferred_solve = (rhs, out, rc_dev, _repairs, hand_over_coupling)
            timer.stop('back_solve')
            return out
        if self._checking():
            before = self.refinement_steps
            bad = self._verify_solution(bc_dev=rc_dev)
            if bad is not None:
                if _repairs > 0 and self._repair_after_inaccurate_solve(bad):
                    timer.stop('back_solve')
                    if self._result_buffers > 0:
                        self._dev_turn -= 1            # (the same result vector again)
                    return self._device_back_solve(rhs, timer, _repairs - 1)
                self._give_up_on_solution(bad)
            if self.refinement_steps != before:
                hand_over_coupling()
        timer.stop('back_solve')
        return out

    def get_inertia(self):
        if self._num_status is None:
            raise RuntimeError('Must call do_numeric_factorization before inertia can be computed')
        if self._general_mode:
            raise RuntimeError('the last matrix was not symmetric: it was factorised through its symmetric embedding '
                               '(general_blocks.py), whose inertia says nothing about it -- the reference counts the eigenvalues '
                               'of such a matrix on the host (scipy_interface.py:39-44), which this package does not do')
        return self._inertia

    def increase_memory_allocation(self, factor):
        self._eng.increase_memory_allocation(factor)

    # ---- general-LU semantics on the ScipyInterface route (general_blocks.py) ----

    def _unsymmetric_anywhere(self, matrix):
        """Collective on the general route: does any rank hold a diagonal block (or the corner) that is not symmetric?"""
        if not self._general_route or hasattr(matrix, 'value_maps') or hasattr(matrix, 'flat_values') or \
                hasattr(matrix, 'diagonal_shift'):
            return False
        last = matrix.bshape[0] - 1
        mine = any(not general_blocks.is_symmetric(matrix.get_block(ndx, ndx)) for ndx in self.local_block_indices) or \
            not general_blocks.is_symmetric(matrix.get_block(last, last))
        if self.comm.size > 1:
            mine = bool(self.comm.allreduce_max(np.array([1 if mine else 0], dtype=np.int64))[0])
        return mine

    def _general_solver(self):
        if self._general_inner is None:
            u0, u1 = self._u_user
            self._general_inner = HipSchurComplementLinearSolver(
                comm=self.comm, engine=self._eng, general_blocks=False, pivot_tolerance=u1 or None,
                symbolic_pivot_threshold=u0 or None)
        return self._general_inner

    def get_schur_complement(self):
        """All-reduced S (without Q) -- parity hook (reference: self.schur_complement): dense array, or for a
        block-tridiagonal S a SciPy COO matrix in the caller's ordering of the coupling variables."""
        if self._general_mode:
            # (general_blocks.py: the embedded system's S is [[0, S], [S', 0]] with S = -sum A K^-1 A^T the reference's and
            # S' the same with K^-T)
            Sbar = self._general_inner.get_schur_complement()
            Sbar = Sbar.toarray() if hasattr(Sbar, 'toarray') else np.asarray(Sbar)
            nc = Sbar.shape[0] // 2
            return np.array(Sbar[:nc, nc:])
        if self._btd is None:
            return self._eng.get_schur()
        from scipy.sparse import coo_matrix
        gs, G = self._btd
        g2 = gs * gs
        flat = self._eng.get_schur_flat()
        D = flat[:G * g2].reshape(G, gs, gs)              # [t][col][row]
        E = flat[G * g2:].reshape(G - 1, gs, gs)
        rows, cols, vals = [], [], []
        t, c, r = np.nonzero(D)
        rows.append(t * gs + r); cols.append(t * gs + c); vals.append(D[t, c, r])
        t, c, r = np.nonzero(E)
        rows += [(t + 1) * gs + r, t * gs + c]
        cols += [t * gs + c, (t + 1) * gs + r]
        vals += [E[t, c, r], E[t, c, r]]
        rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
        keep = (rows < self._nc + 0 * rows) | True
        pr, pc = np.minimum(rows, gs * G - 1), np.minimum(cols, gs * G - 1)
        real = (self._cperm_pad[pr] >= 0) & (self._cperm_pad[pc] >= 0)
        return coo_matrix((vals[real], (self._cperm_pad[pr[real]], self._cperm_pad[pc[real]])), shape=(self._nc, self._nc))

# The serial class of the reference (explicit_schur_complement.py:16) is the same algebra without
# ownership: a BlockMatrix has no rank_ownership, so every block is local.
HipSerialSchurComplementLinearSolver = HipSchurComplementLinearSolver


def __getattr__(name):
    # (the single-matrix adapters live in sub_solvers.py, which imports this module: resolved on first use)
    if name in ('HipLDLInterface', 'MumpsInterface', 'ScipyInterface'):
        from parapint_amd.linalg import sub_solvers
        return getattr(sub_solvers, name)
    raise AttributeError(name)
