"""A-posteriori check of every back-solve, iterative refinement, and repair of a pivot sequence that turns out
inaccurate (mixin of HipSchurComplementLinearSolver; round 6).

The reference's sub-solvers pivot each block on its own values (MA27 with cntl(1): ma27_interface.py:36-47, 110-140;
SuperLU with partial pivoting: scipy_interface.py:26-31), so a `successful` factorisation of theirs solves accurately.
The batched factorisation fixes ONE static pivot sequence per pattern group; a sequence that survives without a zero
pivot can still be unstable for some instance (measured: tools/fuzz_solver.py --hard).  So ``do_back_solve`` never hands
out a solution it has not looked at:

  1. the engine evaluates, on the device, r_i = b_i - K_i x_i - A_i^T x_c for every local block from the values the
     factorisation read, and the row-wise backward error rho = max |r| / max (|K||x| + |A^T x_c| + |b|) of the worst
     instance (csrc/refine.hip).  The coupling rows need no check of their own: S, r_s and the backward sweep use the same
     block factors, so those rows hold to rounding whatever the factors are -- every error of a block factorisation
     shows in the rows of that block;
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

    def _init_solution_check(self):
        self.last_residual = None           # rho of the last back-solve as handed out (max over the ranks)
        self.last_residual_first = None     # ... before refinement
        self.refinement_steps = 0           # correction solves so far
        self.solves_refined = 0             # back-solves that needed at least one
        self.solve_repairs = 0              # back-solves answered by a new pivot sequence + factorisation
        self.inaccurate_solves = 0          # back-solves that stayed above residual_tolerance (raised / warned)
        self._last_factor_call = None       # ('full', matrix) | ('shift', (delta_w, delta_c, coupling_shift, coupling_classes))

    def _checking(self):
        return (self.residual_check and hasattr(self._eng, 'residual') and self._groups is not None and
                self._num_status in (LinearSolverStatus.successful, LinearSolverStatus.warning))

    def _rho(self, store=False):
        """(max over the ranks, this rank's (rho, group, slot))"""
        rho, gid, slot = self._eng.residual(store)
        both = rho
        if self.comm.size > 1:
            both = float(self.comm.allreduce_max(np.array([rho if rho == rho else np.inf], dtype=np.double))[0])
        return both, (rho, gid, slot)

    def _verify_solution(self):
        """After the backward sweep: check, refine.  Returns None if the solution in the engine's vectors is accurate,
        else this rank's (rho, group, slot)."""
        rho, mine = self._rho()
        self.last_residual_first = rho
        steps = 0
        while rho > self.refine_tolerance and np.isfinite(rho) and steps < self.max_refinement_steps:
            if steps == 0:
                _, mine = self._rho(store=True)        # (the same residual once more, kept as the right-hand side)
            self._eng.refine_begin()
            try:
                self._eng.solve_forward()
                self._eng.allreduce_rs(self.comm)
                self._solve_coupling_zero()
                self._eng.solve_backward()
            finally:
                self._eng.refine_end()
            steps += 1
            before = rho
            rho, mine = self._rho(store=True)
            if not rho < 0.5 * before:                 # (stagnation: more of the same does not help)
                break
        if steps:
            self.refinement_steps += steps
            self.solves_refined += 1
        self.last_residual = rho
        if rho <= self.residual_tolerance:
            return None
        return mine

    def _solve_coupling_zero(self):
        dev = getattr(self._eng, 'solve_coupling_dev', None)
        if dev is not None:
            dev(None)
        else:
            self._eng.solve_coupling(None)

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
        if not self._refresh_pivot_order(shift, forced=forced):
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
