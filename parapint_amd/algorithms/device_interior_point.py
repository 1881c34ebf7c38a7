"""The interior-point loop with the iterates resident in HBM.

``ip_solve_device`` is parapint/algorithms/interior_point.py:405-631 (restated for host containers in
``parapint_amd.algorithms.interior_point``) over a device producer
(``parapint_amd.interfaces.schur_complement.device_sc_ip_interface.DeviceStochasticQPInterface``): the KKT values go from
the producer's tensors into the factorisation kernels, the right-hand side and the step are ``DeviceBlockVector``s, and
everything the reference does between two linear solves -- bound-dual steps, fraction to the boundary, the step, the
convergence measures, the next right-hand side -- is a handful of HIP kernels (``csrc/ipstep.hip``).  The
inertia-correction loop is the reference's own (``numeric_factorization`` below is the HOST function, unchanged:
``regularize_*`` return a diagonal shift of the resident matrix and ``do_numeric_factorization`` recognises it).  Per
iteration the host waits twice -- for the factorisation's status and inertia, and for seven scalars of the new iterate:
what the control flow of the reference's loop needs -- and nothing else leaves the device.  With more than one rank the
scenarios are dealt round-robin (mpi_sc_ip_interface.py:14-29) and every rank runs this loop on its own GPU.
"""
import logging
import time

from parapint_amd.algorithms.interior_point import (IPOptions, InteriorPointStatus, _NullTimer, numeric_factorization,
                                                    try_factorization_and_reallocation)
from parapint_amd.linalg.results import LinearSolverStatus

logger = logging.getLogger(__name__)


def ip_solve_device(interface, options=None, timer=None, history=None, stats=None):
    """Returns (status, iterations).  `history`, if a list, receives per iteration
    (primal_inf, dual_inf, compl_inf, barrier, alpha_primal, alpha_dual, regularisation); `stats`, if a dict, the wall
    time of the symbolic phase + set-up (`setup_s`) and of the iterations (`loop_s`)."""
    if options is None:
        options = IPOptions()
    if timer is None:
        timer = _NullTimer()
    solver = options.linalg.solver
    barrier_parameter = options.init_barrier_parameter
    inertia_coef = options.inertia_correction.init_coef
    used_inertia_coef = 0
    t0 = time.time()
    # symbolic phase once, on the matrix of the processed initial point (interior_point.py:542-552)
    dk = interface.device_kkt_matrix()
    sym_status, _ = try_factorization_and_reallocation(dk, solver, options.linalg.reallocation_factor,
                                                       options.linalg.max_num_reallocations, 'symbolic', timer)
    if sym_status != LinearSolverStatus.successful:
        raise RuntimeError('Could not factorize KKT system; linear solver status: ' + str(sym_status))
    interface.attach(solver, dk)
    interface.set_barrier_parameter(barrier_parameter)
    interface.take_step()                                   # (no step yet: barrier diagonals + measures of the initial point)
    m = interface.check_convergence(options.error_scaling)
    t_loop = time.time()
    counter = _torch_op_counter() if stats is not None else None
    alpha_primal_max = alpha_dual_max = 1
    logger.info('%-6s%-11s%-11s%-11s%-11s%-11s%-11s%-11s%-11s%-7s', 'Iter', 'Objective', 'Prim Inf', 'Dual Inf',
                'Comp Inf', 'Barrier', 'Prim Step', 'Dual Step', 'Reg', 'Time')
    status = InteriorPointStatus.error
    iterations = 0
    stamps = [t_loop]
    for _iter in range(options.max_iter):
        iterations = _iter
        primal_inf, dual_inf, compl_inf = m['primal_inf'], m['dual_inf'], m['compl_inf']        # check_convergence(barrier=0)
        if logger.isEnabledFor(logging.INFO):
            logger.info('%-6d%-11.2e%-11.2e%-11.2e%-11.2e%-11.2e%-11.2e%-11.2e%-11.2e%-7.3f', _iter,
                        interface.evaluate_objective(), primal_inf, dual_inf, compl_inf, barrier_parameter,
                        alpha_primal_max, alpha_dual_max, used_inertia_coef, time.time() - t0)
        if history is not None:
            history.append((primal_inf, dual_inf, compl_inf, barrier_parameter, alpha_primal_max, alpha_dual_max,
                            used_inertia_coef))
        if max(primal_inf, dual_inf, compl_inf) <= options.tol:
            status = InteriorPointStatus.optimal
            break
        # check_convergence(barrier=barrier_parameter): the same pass gave the complementarity measure at the barrier
        if max(primal_inf, dual_inf, m['compl_inf_barrier']) <= options.barrier_decrease * barrier_parameter:
            barrier_parameter = max(options.minimum_barrier_parameter,
                                    min(0.5 * barrier_parameter, barrier_parameter ** 1.5))
        interface.set_barrier_parameter(barrier_parameter)
        kkt = interface.evaluate_primal_dual_kkt_matrix(timer=timer)      # (the barrier diagonals are in the solver's sources)
        rhs = interface.evaluate_primal_dual_kkt_rhs(timer=timer)
        if hasattr(solver, 'prefetch_forward'):
            solver.prefetch_forward(rhs)                 # (the forward sweep rides behind every factorisation of this iteration)
        used_inertia_coef = numeric_factorization(interface, kkt, options, inertia_coef, timer)
        inertia_coef = max(used_inertia_coef * options.inertia_correction.factor_decrease,
                           options.inertia_correction.init_coef)
        if hasattr(solver, 'do_back_solve_deferred'):
            # (this package's solver class: the a-posteriori check of the back-solve is enqueued with it, the step lengths --
            # which only READ the solution -- are enqueued behind it, and only then the host waits for the verdict: the
            # stream stays busy meanwhile.  A solution the check had to refine or solve again gets its step lengths again.)
            delta = solver.do_back_solve_deferred(rhs)
            interface.set_primal_dual_kkt_solution(delta)
            interface.fraction_to_the_boundary(1 - barrier_parameter)
            confirmed = solver.confirm_solution()
            if solver.solution_changed_on_confirm:
                delta = confirmed
                interface.set_primal_dual_kkt_solution(delta)
                interface.fraction_to_the_boundary(1 - barrier_parameter)
        else:
            delta = solver.do_back_solve(rhs)
            interface.set_primal_dual_kkt_solution(delta)
            interface.fraction_to_the_boundary(1 - barrier_parameter)
        interface.take_step(unified=options.unified_step)
        m = interface.check_convergence(options.error_scaling)
        alpha_primal_max, alpha_dual_max = m['alpha_primal'], m['alpha_dual']
        if options.unified_step:
            alpha_primal_max = alpha_dual_max = min(alpha_primal_max, alpha_dual_max)
        stamps.append(time.time())
    if stats is not None:
        stats['setup_s'] = t_loop - t0
        stats['loop_s'] = time.time() - t_loop
        stats['iteration_s'] = [b - a for a, b in zip(stamps, stamps[1:])]
        stats['torch_ops'] = counter.close() if counter is not None else None
        stats['torch_op_names'] = dict(counter.names) if counter is not None else None
    return status, iterations


class _torch_op_counter(object):
    """Counts the torch operators dispatched while it is open (diagnostic: the iterations of the loop are meant to run on
    the library's kernels alone; None if torch is not importable -- the CPU tests' numpy engines)."""

    def __init__(self):
        self.count, self._mode, self.names = 0, None, {}
        try:
            from torch.utils._python_dispatch import TorchDispatchMode
        except Exception:
            return
        outer = self

        class _Mode(TorchDispatchMode):
            def __torch_dispatch__(self, func, types, args=(), kwargs=None):
                outer.count += 1
                outer.names[str(func)] = outer.names.get(str(func), 0) + 1
                return func(*args, **(kwargs or {}))
        self._mode = _Mode()
        self._mode.__enter__()

    def close(self):
        if self._mode is None:
            return None
        self._mode.__exit__(None, None, None)
        self._mode = None
        return self.count
