"""The caller of the hot path: parapint's interior-point loop, restated without Pyomo.

``ip_solve`` follows parapint/algorithms/interior_point.py:405-631 step for step -- initial point processing
(:761-799), the two convergence checks per iteration (:174-317), barrier update (:553-557), KKT matrix and
right-hand side from the interface, symbolic factorisation once (:569-580), numeric factorisation inside the
inertia-correction loop (:337-402) behind ``try_factorization_and_reallocation`` (:634-652), back-solve (:566 /
:593), fraction-to-the-boundary rule (:655-758) and the primal / dual step -- and calls the linear solver only
through the ``LinearSolverInterface`` surface with the reference's keywords (``matrix=kkt, raise_on_error=False,
timer=timer``).  It exists so that the solver can be driven the way parapint drives it (SURVEY.md section 8 rows a20,
C1, f2, f4); it is not part of the solver.

Vectors are kept as flat local arrays with the block layout of the interface's containers; for containers that
carry an ownership table the reductions (max norms, sums, step lengths) go through the interface's communicator,
which is what PyNumero's MPIBlockVector does implicitly in the reference.
"""
import enum
import logging
import time

import numpy as np

from parapint_amd.linalg.results import LinearSolverStatus

logger = logging.getLogger(__name__)


class InteriorPointStatus(enum.Enum):
    optimal = 0
    error = 1


class _Options(object):
    def __setattr__(self, name, value):
        if not hasattr(self, name) and not name.startswith('_'):
            raise AttributeError('unknown option ' + name)
        object.__setattr__(self, name, value)


class InertiaCorrectionOptions(_Options):      # interior_point.py:30-58
    init_coef = 1e-8
    factor_increase = 10
    factor_decrease = 1 / 3
    max_coef = 1e9


class LinalgOptions(_Options):                 # :61-86
    solver = None
    reallocation_factor = 2
    max_num_reallocations = 5


class LineSearchOptions(_Options):             # :89-114 (the reference's line search is a placeholder)
    max_iter = 4
    disable = True
    step_anyway = True


class IPOptions(_Options):                     # :117-171
    def __init__(self):
        object.__setattr__(self, 'inertia_correction', InertiaCorrectionOptions())
        object.__setattr__(self, 'linalg', LinalgOptions())
        object.__setattr__(self, 'line_search', LineSearchOptions())

    max_iter = 1000
    tol = 1e-8
    init_barrier_parameter = 0.1
    minimum_barrier_parameter = 1e-9
    barrier_decrease = 10
    report_timing = False
    use_inertia_correction = True
    inertia_correction = None
    linalg = None
    line_search = None
    unified_step = False
    error_scaling = 100
    bounds_relaxation_factor = 1e-8


class _NullTimer(object):
    def start(self, name):
        pass

    def stop(self, name):
        pass


class _Layout(object):
    """Flat view of a (possibly nested, possibly rank-distributed) block vector."""

    def __init__(self, template, comm):
        self.template = template
        self.comm = comm
        self.plain = not hasattr(template, 'nblocks')         # a single problem's interface hands plain arrays
        if self.plain:
            self.size = int(np.asarray(template).size)
            self.slices = []
            self.counted = np.ones(self.size, dtype=bool)
            return
        owner = getattr(template, 'rank_ownership', None)
        self.slices = []
        off = 0
        counted = []
        for i in range(template.nblocks):
            blk = template.get_block(i)
            if blk is None:
                continue
            size = blk.size
            self.slices.append((i, off, off + size))
            replicated = owner is not None and owner[i] == -1
            counted.append(np.full(size, (not replicated) or comm is None or comm.rank == 0))
            off += size
        self.size = off
        self.counted = np.concatenate(counted) if counted else np.zeros(0, dtype=bool)

    def flat(self, bv):
        if self.plain:
            return np.array(bv, dtype=np.double).ravel()
        if self.size == 0:
            return np.zeros(0)
        parts = []
        for i, a, b in self.slices:
            blk = bv.get_block(i)
            parts.append(blk.flatten() if hasattr(blk, 'get_block') else np.asarray(blk, dtype=np.double).ravel())
        return np.concatenate(parts).astype(np.double, copy=True)

    def unflat(self, arr):
        if self.plain:
            return np.array(arr, dtype=np.double)
        t = self.template
        out = t.copy_structure_unset() if hasattr(t, 'copy_structure_unset') else t.copy_structure()
        for i, a, b in self.slices:
            blk = t.get_block(i)
            if hasattr(blk, 'get_block'):
                sub = blk.copy_structure()
                sub.copyfrom(arr[a:b])
                out.set_block(i, sub)
            else:
                out.set_block(i, arr[a:b].copy())
        return out


class _Reduce(object):
    def __init__(self, comm):
        self.comm = comm if (comm is not None and comm.size > 1) else None

    def max(self, v):
        v = float(v)
        return v if self.comm is None else float(self.comm.allreduce_max(np.array([v]))[0])

    def min(self, v):
        return -self.max(-float(v))

    def sum(self, v):
        v = float(v)
        return v if self.comm is None else float(self.comm.allreduce_sum(np.array([v]))[0])


def _max_abs(x):
    return float(np.max(np.abs(x))) if x.size else 0.0


def check_convergence(interface, barrier, error_scaling, state, lay, red, timer=None):
    """interior_point.py:174-317: (primal infeasibility, scaled dual infeasibility, scaled complementarity)."""
    (primals, slacks, duals_eq, duals_ineq, zl, zu, sl, su) = state
    grad_obj = interface.get_obj_factor() * lay['primals'].flat(interface.evaluate_grad_objective())
    eq_resid = lay['eq'].flat(interface.evaluate_eq_constraints())
    ineq_resid = lay['ineq'].flat(interface.evaluate_ineq_constraints()) - slacks
    jt = lay['primals'].flat(interface.grad_lag_primals_terms())
    plb, pub = lay['primals'].flat(interface.primals_lb()), lay['primals'].flat(interface.primals_ub())
    ilb, iub = lay['ineq'].flat(interface.ineq_lb()), lay['ineq'].flat(interface.ineq_ub())
    grad_lag_primals = grad_obj + jt - zl + zu
    grad_lag_slacks = -duals_ineq - sl + su

    def bound_resid(x, lb, dual, lower):
        mod = np.where(np.isfinite(lb), lb, 0.0)
        r = ((x - mod) if lower else (mod - x)) * dual - barrier
        r[~np.isfinite(lb)] = 0
        return r
    compl = max(_max_abs(bound_resid(primals, plb, zl, True)), _max_abs(bound_resid(primals, pub, zu, False)),
                _max_abs(bound_resid(slacks, ilb, sl, True)), _max_abs(bound_resid(slacks, iub, su, False)))
    primal_inf = red.max(max(_max_abs(eq_resid), _max_abs(ineq_resid)))
    dual_inf = red.max(max(_max_abs(grad_lag_primals), _max_abs(grad_lag_slacks)))
    compl = red.max(compl)
    cp, ci, ce = lay['primals'].counted, lay['ineq'].counted, lay['eq'].counted
    bound_sum = (np.abs(zl[cp]).sum() + np.abs(zu[cp]).sum() + np.abs(sl[ci]).sum() + np.abs(su[ci]).sum())
    n_bounds = (np.isfinite(plb[cp]).sum() + np.isfinite(pub[cp]).sum() + np.isfinite(ilb[ci]).sum() +
                np.isfinite(iub[ci]).sum())
    dual_sum = red.sum(np.abs(duals_eq[ce]).sum() + np.abs(duals_ineq[ci]).sum() + bound_sum)
    dual_cnt = red.sum(ce.sum() + ci.sum() + n_bounds)
    bound_sum, n_bounds = red.sum(bound_sum), red.sum(n_bounds)
    dual_scaling = max(error_scaling, dual_sum / dual_cnt) / error_scaling
    compl_scaling = max(error_scaling, bound_sum / n_bounds) / error_scaling if n_bounds > 0 else 1.0
    return primal_inf, dual_inf / dual_scaling, compl / compl_scaling


def try_factorization_and_reallocation(kkt, linear_solver, reallocation_factor, max_iter, symbolic_or_numeric, timer=None):
    """interior_point.py:634-652."""
    assert max_iter >= 1
    method = linear_solver.do_numeric_factorization if symbolic_or_numeric == 'numeric' else \
        linear_solver.do_symbolic_factorization
    for count in range(max_iter):
        res = method(matrix=kkt, raise_on_error=False, timer=timer)
        status = res.status
        if status == LinearSolverStatus.not_enough_memory:
            linear_solver.increase_memory_allocation(reallocation_factor)
        else:
            break
    return status, count


def numeric_factorization(interface, kkt, options, inertia_coef, timer=None):
    """interior_point.py:337-402: factorise; while the inertia is wrong, regularise and factorise again."""
    solver = options.linalg.solver
    status, _ = try_factorization_and_reallocation(kkt, solver, options.linalg.reallocation_factor,
                                                   options.linalg.max_num_reallocations, 'numeric', timer)
    final_inertia_coef = 0
    if not options.use_inertia_correction:
        if status != LinearSolverStatus.successful:
            raise RuntimeError('Could not factorize KKT system; linear solver status: ' + str(status))
        return final_inertia_coef
    if status not in {LinearSolverStatus.successful, LinearSolverStatus.singular}:
        raise RuntimeError('Could not factorize KKT system; linear solver status: ' + str(status))
    n_con = interface.n_eq_constraints() + interface.n_ineq_constraints()
    neg_eig = zero_eig = None
    _iter = 0
    while final_inertia_coef <= options.inertia_correction.max_coef:
        if status == LinearSolverStatus.successful:
            _, neg_eig, zero_eig = solver.get_inertia()
        else:
            neg_eig = zero_eig = None
        logger.debug('reg_iter %d reg_coef %.2e neg_eig %s zero_eig %s status %s', _iter, final_inertia_coef, neg_eig,
                     zero_eig, status)
        if neg_eig == n_con and zero_eig == 0 and status == LinearSolverStatus.successful:
            break
        if _iter == 0:
            kkt = kkt.copy()
        kkt = interface.regularize_equality_gradient(kkt=kkt, coef=-inertia_coef, copy_kkt=False)
        kkt = interface.regularize_hessian(kkt=kkt, coef=inertia_coef, copy_kkt=False)
        status, _ = try_factorization_and_reallocation(kkt, solver, options.linalg.reallocation_factor,
                                                       options.linalg.max_num_reallocations, 'numeric', timer)
        final_inertia_coef = inertia_coef
        inertia_coef *= options.inertia_correction.factor_increase
        _iter += 1
    if neg_eig != n_con or zero_eig != 0 or status != LinearSolverStatus.successful:
        raise RuntimeError('Exceeded maximum inertia correciton')
    return final_inertia_coef


def process_init(x, lb, ub):
    """interior_point.py:761-788: push an initial point inside its bounds."""
    if np.any((ub - lb) < 0):
        raise ValueError('Lower bounds for variables/inequalities should not be larger than upper bounds.')
    if np.any((ub - lb) == 0):
        raise ValueError('Variables and inequalities should not have equal lower and upper bounds.')
    has_lb, has_ub = np.isfinite(lb), np.isfinite(ub)
    out = (x >= ub) | (x <= lb)
    only_lb = out & has_lb & ~has_ub
    only_ub = out & has_ub & ~has_lb
    both = out & has_lb & has_ub
    x[only_lb] = lb[only_lb] + 1
    x[only_ub] = ub[only_ub] - 1
    x[both] = 0.5 * (lb[both] + ub[both])


def process_init_duals_lb(x, lb):
    x[x <= 0] = 1
    x[np.isneginf(lb)] = 0


def process_init_duals_ub(x, ub):
    x[x <= 0] = 1
    x[np.isinf(ub)] = 0


def _frac_lb(tau, x, dx, xl):
    """interior_point.py:655-663 (and, with x -> -x, :666-674)."""
    if x.size == 0:
        return 1.0
    mod = np.where(dx == 0, 1.0, dx)
    with np.errstate(invalid='ignore', over='ignore'):
        alpha = -tau * (x - xl) / mod
    alpha[dx >= 0] = np.inf
    return min(float(alpha.min()), 1.0)


def _frac_ub(tau, x, dx, xu):
    if x.size == 0:
        return 1.0
    mod = np.where(dx == 0, 1.0, dx)
    with np.errstate(invalid='ignore', over='ignore'):
        alpha = tau * (xu - x) / mod
    alpha[dx <= 0] = np.inf
    return min(float(alpha.min()), 1.0)


def fraction_to_the_boundary(tau, state, deltas, bounds, red):
    """interior_point.py:677-758."""
    (primals, slacks, _, _, zl, zu, sl, su) = state
    (dp, ds, _, _, dzl, dzu, dsl, dsu) = deltas
    plb, pub, ilb, iub = bounds
    a_p = min(_frac_lb(tau, primals, dp, plb), _frac_ub(tau, primals, dp, pub),
              _frac_lb(tau, slacks, ds, ilb), _frac_ub(tau, slacks, ds, iub))
    a_d = min(_frac_lb(tau, zl, dzl, np.zeros_like(zl)), _frac_lb(tau, zu, dzu, np.zeros_like(zu)),
              _frac_lb(tau, sl, dsl, np.zeros_like(sl)), _frac_lb(tau, su, dsu, np.zeros_like(su)))
    return red.min(a_p), red.min(a_d)


def ip_solve(interface, options=None, timer=None):
    """interior_point.py:405-631."""
    if options is None:
        options = IPOptions()
    if timer is None:
        timer = _NullTimer()
    comm = getattr(interface, '_comm', None)
    red = _Reduce(comm)
    solver = options.linalg.solver
    timer.start('IP solve')
    timer.start('init')
    interface.set_bounds_relaxation_factor(options.bounds_relaxation_factor)
    barrier_parameter = options.init_barrier_parameter
    inertia_coef = options.inertia_correction.init_coef
    used_inertia_coef = 0
    t0 = time.time()
    lay = {'primals': _Layout(interface.init_primals(), comm), 'ineq': _Layout(interface.init_slacks(), comm),
           'eq': _Layout(interface.init_duals_eq(), comm)}
    P, I, E = lay['primals'], lay['ineq'], lay['eq']
    primals = P.flat(interface.init_primals())
    slacks = I.flat(interface.init_slacks())
    duals_eq = E.flat(interface.init_duals_eq())
    duals_ineq = I.flat(interface.init_duals_ineq())
    zl, zu = P.flat(interface.init_duals_primals_lb()), P.flat(interface.init_duals_primals_ub())
    sl, su = I.flat(interface.init_duals_slacks_lb()), I.flat(interface.init_duals_slacks_ub())
    plb, pub = P.flat(interface.primals_lb()), P.flat(interface.primals_ub())
    ilb, iub = I.flat(interface.ineq_lb()), I.flat(interface.ineq_ub())
    process_init(primals, plb, pub)
    process_init(slacks, ilb, iub)
    process_init_duals_lb(zl, plb)
    process_init_duals_ub(zu, pub)
    process_init_duals_lb(sl, ilb)
    process_init_duals_ub(su, iub)
    interface.set_barrier_parameter(barrier_parameter)
    alpha_primal_max = alpha_dual_max = alpha = 1
    logger.info('%-6s%-11s%-11s%-11s%-11s%-11s%-11s%-11s%-11s%-11s%-7s', 'Iter', 'Objective', 'Prim Inf', 'Dual Inf',
                'Comp Inf', 'Barrier', 'Prim Step', 'Dual Step', 'LS Step', 'Reg', 'Time')
    timer.stop('init')
    status = InteriorPointStatus.error
    for _iter in range(options.max_iter):
        interface.set_primals(P.unflat(primals))
        interface.set_slacks(I.unflat(slacks))
        interface.set_duals_eq(E.unflat(duals_eq))
        interface.set_duals_ineq(I.unflat(duals_ineq))
        interface.set_duals_primals_lb(P.unflat(zl))
        interface.set_duals_primals_ub(P.unflat(zu))
        interface.set_duals_slacks_lb(I.unflat(sl))
        interface.set_duals_slacks_ub(I.unflat(su))
        state = (primals, slacks, duals_eq, duals_ineq, zl, zu, sl, su)
        timer.start('convergence check')
        primal_inf, dual_inf, compl_inf = check_convergence(interface, 0, options.error_scaling, state, lay, red, timer)
        timer.stop('convergence check')
        objective = interface.evaluate_objective()
        logger.info('%-6d%-11.2e%-11.2e%-11.2e%-11.2e%-11.2e%-11.2e%-11.2e%-11.2e%-11.2e%-7.3f', _iter, objective,
                    primal_inf, dual_inf, compl_inf, barrier_parameter, alpha_primal_max, alpha_dual_max, alpha,
                    used_inertia_coef, time.time() - t0)
        if max(primal_inf, dual_inf, compl_inf) <= options.tol:
            status = InteriorPointStatus.optimal
            break
        timer.start('convergence check')
        primal_inf, dual_inf, compl_inf = check_convergence(interface, barrier_parameter, options.error_scaling, state,
                                                            lay, red, timer)
        timer.stop('convergence check')
        if max(primal_inf, dual_inf, compl_inf) <= options.barrier_decrease * barrier_parameter:
            barrier_parameter = max(options.minimum_barrier_parameter,
                                    min(0.5 * barrier_parameter, barrier_parameter ** 1.5))
        interface.set_barrier_parameter(barrier_parameter)
        timer.start('eval')
        timer.start('eval kkt')
        kkt = interface.evaluate_primal_dual_kkt_matrix(timer=timer)
        timer.stop('eval kkt')
        timer.start('eval rhs')
        rhs = interface.evaluate_primal_dual_kkt_rhs(timer=timer)
        timer.stop('eval rhs')
        timer.stop('eval')
        timer.start('factorize')
        if _iter == 0:
            timer.start('symbolic')
            sym_status, _ = try_factorization_and_reallocation(kkt, solver, options.linalg.reallocation_factor,
                                                               options.linalg.max_num_reallocations, 'symbolic', timer)
            timer.stop('symbolic')
            if sym_status != LinearSolverStatus.successful:
                raise RuntimeError('Could not factorize KKT system; linear solver status: ' + str(sym_status))
        timer.start('numeric')
        if hasattr(solver, 'prefetch_forward') and hasattr(rhs, 'group_tensors'):
            # (extension of this package's solver class, a no-op for every other LinearSolverInterface: the right-hand side
            # of this iteration exists before the matrix is factorised -- interior_point.py:553-566 -- so a device-resident
            # one is announced and its forward sweep runs beside the dense factorisation of S)
            solver.prefetch_forward(rhs)
        try:
            used_inertia_coef = numeric_factorization(interface, kkt, options, inertia_coef, timer)
        except Exception:
            # (the announcement ends with the back-solve; a factorisation that raises never gets there -- withdraw it, or
            # whoever reuses the solver sweeps a stale vector behind its next factorisation)
            if hasattr(solver, 'prefetch_forward'):
                solver.prefetch_forward(None)
            raise
        inertia_coef = max(used_inertia_coef * options.inertia_correction.factor_decrease,
                           options.inertia_correction.init_coef)
        timer.stop('numeric')
        timer.stop('factorize')
        timer.start('back solve')
        delta = solver.do_back_solve(rhs)
        timer.stop('back solve')
        interface.set_primal_dual_kkt_solution(delta)
        timer.start('frac boundary')
        deltas = (P.flat(interface.get_delta_primals()), I.flat(interface.get_delta_slacks()),
                  E.flat(interface.get_delta_duals_eq()), I.flat(interface.get_delta_duals_ineq()),
                  P.flat(interface.get_delta_duals_primals_lb()), P.flat(interface.get_delta_duals_primals_ub()),
                  I.flat(interface.get_delta_duals_slacks_lb()), I.flat(interface.get_delta_duals_slacks_ub()))
        plb, pub = P.flat(interface.primals_lb()), P.flat(interface.primals_ub())
        ilb, iub = I.flat(interface.ineq_lb()), I.flat(interface.ineq_ub())
        alpha_primal_max, alpha_dual_max = fraction_to_the_boundary(1 - barrier_parameter, state, deltas,
                                                                    (plb, pub, ilb, iub), red)
        if options.unified_step:
            alpha_primal_max = alpha_dual_max = min(alpha_primal_max, alpha_dual_max)
        timer.stop('frac boundary')
        alpha = 1                                       # (line search disabled: the reference's is a placeholder)
        primals += alpha * alpha_primal_max * deltas[0]
        slacks += alpha * alpha_primal_max * deltas[1]
        duals_eq += alpha * alpha_dual_max * deltas[2]
        duals_ineq += alpha * alpha_dual_max * deltas[3]
        zl += alpha * alpha_dual_max * deltas[4]
        zu += alpha * alpha_dual_max * deltas[5]
        sl += alpha * alpha_dual_max * deltas[6]
        su += alpha * alpha_dual_max * deltas[7]
    timer.stop('IP solve')
    return status
