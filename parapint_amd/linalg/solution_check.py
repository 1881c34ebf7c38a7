"""A-posteriori check of every back-solve, iterative refinement, and repair of a pivot sequence that turns out
inaccurate (mixin of HipSchurComplementLinearSolver; round 6).

The reference's sub-solvers pivot each block on its own values (MA27 with cntl(1): ma27_interface.py:36-47, 110-140;
SuperLU with partial pivoting: scipy_interface.py:26-31), so a `successful` factorisation of theirs solves accurately.
The batched factorisation fixes ONE static pivot sequence per pattern group; a sequence that survives without a zero
pivot can still be unstable for some instance (measured: tools/fuzz_solver.py --hard).  So ``do_back_solve`` never hands
out a solution it has not looked at:

  1. the engine evaluates, on the device, r_i = b_i - K_i x_i - A_i^T x_c for every local block from the values the
     factorisation read, and the row-wise backward error rho = max |r| / max (|K||x| + |A^T x_c| + |b|) of the worst
     instance (csrc/refine.hip), and the sums sum_i A_i x_i, sum_i |A_i||x_i| of the coupling rows, which this class
     finishes with Q x_c (n_c-vectors).  In exact arithmetic the coupling rows hold whatever the block factors are (S, r_s and
     the backward sweep use the same factors), but an unstable pivot sequence amplifies their rounding errors;
  2. rho > ``refine_tolerance`` (1e-10): up to ``max_refinement_steps`` correction solves K d = r through the same sweeps
     (one more all-reduce of r_s each), x += d;
  3. still rho > ``residual_tolerance`` (1e-8, the bound BASELINE.json's north_star states): the pivot sequence of the
     group that holds the worst instance is chosen again from THAT instance's values (the refresh / group-splitting path
     of pivot_repair.py), the matrix is factorised again and the solve repeated;
  4. if that does not help either the solve ends with a RuntimeError (``on_inaccurate_solve = 'raise'``, the default) or
     a logged warning ('warn') -- the status of the factorisation that produced it is then set to `warning`.
With several ranks the decision is taken on the maximum of rho over the ranks (one scalar all-reduce per back-solve):
a correction solve and a new factorisation are collective."""
import numpy as np

from parapint_amd.linalg.results import LinearSolverStatus


class SolutionCheckMixin(object):
    residual_check = True           # False: hand out x unchecked (a caller that verifies its own solutions)
    refine_tolerance = 1e-10
    residual_tolerance = 1e-8
    max_refinement_steps = 2
    max_solve_repairs = 2           # new pivot sequences + factorisations one back-solve may ask for
    on_inaccurate_solve = 'raise'   # or 'warn'
    repair_thresholds = (0.1, 0.3)  # threshold u of the static 1x1 / 2x2 choice for the 1st, 2nd, ... repair of one back-solve

    def _init_solution_check(self):
        self.last_residual = None           # rho of the last back-solve as handed out (max over the ranks)
        self.last_residual_first = None     # ... before refinement
        self.refinement_steps = 0           # correction solves so far
        self.solves_refined = 0             # back-solves that needed at least one
        self.solve_repairs = 0              # back-solves answered by a new pivot sequence + factorisation
        self.inaccurate_solves = 0          # back-solves that stayed above residual_tolerance (raised / warned)
        self._last_factor_call = None       # ('full', matrix) | ('shift', (delta_w, delta_c, coupling_shift, coupling_classes))
        self._check_bc_host = self._check_bc_dev = self._check_rc = None
        self._repairs_this_solve = 0

    def _checking(self):
        return (self.residual_check and hasattr(self._eng, 'residual_begin') and self._groups is not None and
                self._num_status in (LinearSolverStatus.successful, LinearSolverStatus.warning))

    def _rho_begin(self, store=False):
        """Enqueues the check behind the back-solve (no wait): _rho(begun=True) collects it."""
        on_device = 0
        direct = getattr(self._eng, '_direct_rccl', None)
        if direct is not None and (self.comm.size > 1 or getattr(self.comm, 'always_reduce', False)) and direct(self.comm):
            on_device = 2        # (the library's communicator: one all-reduce on its stream, no host collective)
        elif self.comm.size == 1:
            on_device = 1
        self._eng.residual_begin(store, self._check_bc_dev, on_device)

    def _rho(self, store=False, begun=False):
        """Backward error of the solution in the engine's vectors: (max over the block rows of all ranks and over the
        coupling rows, this rank's (rho of its worst block, group, slot)).  The coupling rows b_c - sum_i A_i x_i - Q x_c are
        finished here from the sums the engine formed (n_c-vectors; with several ranks one sum all-reduce carries them and --
        one slot per rank -- the block results, so every rank takes the same decision)."""
        if not begun:
            self._rho_begin(store)
        rb, gid, slot, scale, rho_c, xc, ax, aabs, bcd = self._eng.residual_end()
        if not rb == rb:
            rb = np.inf
        self._check_rc = None
        if rho_c is not None:                     # (the engine judged the coupling rows as well; their residual stays with it)
            self._check_rc = 'engine'
            return max(rb, rho_c if rho_c == rho_c else np.inf), (rb, gid, slot)
        rho_blocks = rb
        if self.comm.size > 1:
            mine = np.zeros(2 * self.comm.size)
            mine[2 * self.comm.rank] = min(rb, 1e300)
            mine[2 * self.comm.rank + 1] = min(scale, 1e300) if scale == scale else 1e300
            buf = self.comm.allreduce_sum(np.concatenate([ax, aabs, mine]))
            n = ax.size
            ax, aabs, rho_blocks, scale = buf[:n], buf[n:2 * n], float(buf[2 * n::2].max()), float(buf[2 * n + 1::2].max())
        rho = rho_blocks
        if self._nc > 0:
            bc = self._check_bc_host if self._check_bc_host is not None else bcd
            if self._btd is not None:        # (library order, padded -> the caller's order, in which Q is kept)
                xc, ax, aabs, bc = xc[self._cinv], ax[self._cinv], aabs[self._cinv], bc[self._cinv]
            Q = self._last_Q
            rc = bc - ax
            sc = np.abs(bc) + aabs
            if Q is not None:
                rc = rc - Q.dot(xc)
                sc = sc + abs(Q).dot(np.abs(xc))
            rc, sc = np.asarray(rc).ravel(), np.asarray(sc).ravel()
            rmax = np.abs(rc).max()
            # (normwise over the whole system: a coupling row whose own terms are all tiny -- its solution component is
            # zero -- is measured against the largest row scale of the blocks, not against itself)
            smax = max(float(sc.max()), scale)
            rho_c = 0.0 if rmax == 0.0 else (rmax / smax if smax > 0.0 and np.isfinite(rmax) else np.inf)
            if not rho_c == rho_c:
                rho_c = np.inf
            rho = max(rho, rho_c)
            self._check_rc = rc
        return rho, (rb, gid, slot)

    def _verify_solution(self, bc_host=None, bc_dev=None, begun=False):
        """After the backward sweep: check, refine.  bc_host / bc_dev: the coupling right-hand side of the back-solve in the
        library's order (numpy array / device tensor; both None: zero).  Returns None if the solution in the engine's
        vectors is accurate, else this rank's (rho of its worst block, group, slot)."""
        self._check_bc_host, self._check_bc_dev = bc_host, bc_dev
        steps = 0
        try:
            rho, mine = self._rho(begun=begun)
            self.last_residual_first = rho
            while rho > self.refine_tolerance and np.isfinite(rho) and steps < self.max_refinement_steps:
                if steps == 0:
                    rho, mine = self._rho(store=True)        # (the same residual once more, kept as the right-hand side)
                rc = self._check_rc
                self._eng.refine_begin()
                try:
                    self._eng.solve_forward()
                    self._eng.allreduce_rs(self.comm)
                    if isinstance(rc, str):
                        self._eng.refine_solve_coupling()        # (the residual of the coupling rows is on the device)
                    else:
                        self._eng.solve_coupling(None if rc is None else self._to_coupling_order(rc))
                    self._eng.solve_backward()
                finally:
                    self._eng.refine_end()
                steps += 1
                before = rho
                rho, mine = self._rho(store=True)
                if not rho < 0.5 * before:                 # (stagnation: more of the same does not help)
                    break
        finally:
            self._check_bc_host = self._check_bc_dev = None
        if steps:
            self.refinement_steps += steps
            self.solves_refined += 1
        self.last_residual = rho
        if rho <= self.residual_tolerance:
            return None
        return mine

    def _repair_after_inaccurate_solve(self, mine):
        """Collective.  New pivot sequence(s) from the worst instance(s), factorise again.  Returns True if a new, successful
        factorisation is in place."""
        last = self._last_factor_call
        if last is None or self.max_solve_repairs <= 0:
            return False
        forced = {}
        if mine is not None and mine[1] >= 0 and not (mine[0] <= self.residual_tolerance):
            forced[int(mine[1])] = int(mine[2])
        shift = (last[1][0], last[1][1]) if last[0] == 'shift' else None
        # the sequence that failed was chosen with the threshold in force: the next one takes more 2 x 2 pivots (MA27 users
        # raise cntl(1) when a factorisation turns out fragile; Ipopt: ma27_pivtol -> ma27_pivtolmax)
        u_next = self.repair_thresholds[min(self._repairs_this_solve, len(self.repair_thresholds) - 1)]
        self._repairs_this_solve += 1
        if not self._refresh_pivot_order(shift, forced=forced, u_min=u_next):
            return False
        self.solve_repairs += 1
        if last[0] == 'full':
            res = self._numeric_factorization(last[1])
        else:
            dw, dc, cs, cc = last[1]
            if self._last_device_base is not None and self._device_maps is not None:
                self._bind_device_matrix(self._last_device_base)
            else:
                for g in self._groups:
                    self._eng.upload_values_compact(g.gid, g.staging)
            res = self.refactorize_with_diagonal_shift(dw, dc, coupling_shift=cs, raise_on_error=False, _retry=True,
                                                       coupling_classes=cc)
        self._note_refresh_outcome(res.status in (LinearSolverStatus.successful, LinearSolverStatus.warning))
        return res.status in (LinearSolverStatus.successful, LinearSolverStatus.warning)

    def _give_up_on_solution(self, mine):
        self.inaccurate_solves += 1
        where = ''
        if mine is not None and mine[1] >= 0 and mine[1] < len(self._groups) and 0 <= mine[2] < len(self._groups[mine[1]].blocks):
            where = ' (worst on this rank: block %d, %.2e)' % (self._groups[mine[1]].blocks[mine[2]], mine[0])
        msg = ('back-solve inaccurate: row-wise backward error %.2e > %.0e after %d refinement step(s) and %d new pivot '
               'sequence(s)%s' % (self.last_residual, self.residual_tolerance, self.max_refinement_steps, self.solve_repairs, where))
        self._num_status = LinearSolverStatus.warning
        if self.on_inaccurate_solve == 'raise':
            raise RuntimeError(msg)
        self._last_error = msg
        import logging
        logging.getLogger(self.getLoggerName()).warning(msg)
