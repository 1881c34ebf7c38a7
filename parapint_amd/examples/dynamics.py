"""The reference's dynamic-optimisation example (parapint/examples/dynamics.py:14-230), Pyomo-free:

    min  integral from t0 to tf of [x(t) - sin(time_scale t) - 1]^2     s.t.  dx/dt = p(t) - x(t),  p(t) <= 2

trapezoid rule for the integral, implicit Euler for the differential equation, the control p piecewise constant over
``constant_control_duration``, the horizon cut into time blocks tied at the state x.  A time block is a QP, stated here as
a ``QuadraticProgram`` where the reference builds a Pyomo block (``build_time_block``, :37-101); ``Problem`` and ``main``
keep the reference's constructor arguments, defaults and call sequence (:104-180).  The reference's own test holds the
optimal controls of the default problem to 7 places (``examples/tests/test_examples.py:38-58``): the known answers
this restatement of the interface and the loop is pinned against (tests/test_dynamics_example.py)."""
import math

import numpy as np
from scipy.sparse import coo_matrix

from parapint_amd.algorithms.interior_point import IPOptions, InteriorPointStatus, ip_solve
from parapint_amd.interfaces.interface import QuadraticProgram
from parapint_amd.interfaces.schur_complement.sc_ip_interface import MPIDynamicSchurComplementInteriorPointInterface


def build_time_block(t0, delta_t, num_finite_elements, constant_control_duration, time_scale, p_ub=2.0, p_ub_as_constraint=None):
    """dynamics.py:37-101.  Returns (QuadraticProgram, x time points, p time points): the variables are x at the
    num_finite_elements + 1 time points, then p at its time points.  p_ub: the bound on the control (None: free);
    p_ub_as_constraint: the same limit stated as inequality constraints p[t] <= value instead (the variants of the
    reference's interface tests, interfaces/schur_complement/tests/test_sc_ip_interface.py:26-99)."""
    assert constant_control_duration >= delta_t
    assert constant_control_duration % delta_t == 0
    assert (num_finite_elements * delta_t) % constant_control_duration == 0
    x_times = [t for t in range(t0, t0 + delta_t * (num_finite_elements + 1), delta_t)]
    num_p_elements = int((num_finite_elements * delta_t) / constant_control_duration)
    p_times = [t for t in range(t0, t0 + constant_control_duration * num_p_elements, constant_control_duration)]
    nx, n = len(x_times), len(x_times) + len(p_times)
    x_at = {t: i for i, t in enumerate(x_times)}
    p_at = {t: nx + i for i, t in enumerate(p_times)}
    hd, c, c0 = np.zeros(n), np.zeros(n), 0.0
    rows, cols, vals = [], [], []
    for fe in range(num_finite_elements):
        start_t_x, end_t_x = t0 + fe * delta_t, t0 + (fe + 1) * delta_t
        start_t_p = t0 + (math.floor(fe / (constant_control_duration / delta_t))) * constant_control_duration
        for t in (start_t_x, end_t_x):                       # 0.5 delta_t (x[t] - (sin(time_scale t) + 1))^2
            w, a = 0.5 * delta_t, math.sin(time_scale * t) + 1
            hd[x_at[t]] += 2.0 * w
            c[x_at[t]] += -2.0 * w * a
            c0 += w * a * a
        # x[end] - (x[start] + delta_t (p[start_p] - x[end])) == 0
        rows += [fe, fe, fe]
        cols += [x_at[end_t_x], x_at[start_t_x], p_at[start_t_p]]
        vals += [1.0 + delta_t, -1.0, -float(delta_t)]
    ub = np.full(n, np.inf)
    if p_ub is not None:
        ub[nx:] = p_ub                                       # bnds = (None, 2)
    A_ineq = ineq_ub = None
    if p_ub_as_constraint is not None:
        k = np.arange(len(p_times))
        A_ineq = coo_matrix((np.ones(k.size), (k, nx + k)), shape=(k.size, n))
        ineq_ub = np.full(k.size, float(p_ub_as_constraint))
    qp = QuadraticProgram(c=c, A_eq=coo_matrix((vals, (rows, cols)), shape=(num_finite_elements, n)),
                          b_eq=np.zeros(num_finite_elements), ub=ub, A_ineq=A_ineq, ineq_ub=ineq_ub,
                          H=coo_matrix((hd, (np.arange(n), np.arange(n))), shape=(n, n)), c0=c0)
    return qp, x_times, p_times


class Problem(MPIDynamicSchurComplementInteriorPointInterface):
    def __init__(self, t0=0, delta_t=1, num_finite_elements=90, constant_control_duration=10, time_scale=0.1,
                 num_time_blocks=3, comm=None, p_ub=2.0, p_ub_as_constraint=None):
        self.p_ub, self.p_ub_as_constraint = p_ub, p_ub_as_constraint
        assert num_finite_elements % num_time_blocks == 0
        self.t0, self.delta_t, self.num_finite_elements = t0, delta_t, num_finite_elements
        self.constant_control_duration, self.time_scale, self.num_time_blocks = constant_control_duration, time_scale, num_time_blocks
        self.tf = self.t0 + self.delta_t * self.num_finite_elements
        self.time_points = {}
        super(Problem, self).__init__(start_t=self.t0, end_t=self.tf, num_time_blocks=self.num_time_blocks, comm=comm)

    def build_model_for_time_block(self, ndx, start_t, end_t, add_init_conditions):
        assert int(start_t) == start_t
        assert int(end_t) == end_t
        assert end_t == start_t + self.delta_t * (self.num_finite_elements / self.num_time_blocks)
        start_t, end_t = int(start_t), int(end_t)
        qp, x_times, p_times = build_time_block(t0=start_t, delta_t=self.delta_t,
                                                num_finite_elements=int(self.num_finite_elements / self.num_time_blocks),
                                                constant_control_duration=self.constant_control_duration,
                                                time_scale=self.time_scale, p_ub=self.p_ub,
                                                p_ub_as_constraint=self.p_ub_as_constraint)
        self.time_points[ndx] = (x_times, p_times)
        return qp, [x_times.index(start_t)], [x_times.index(end_t)]

    def x(self, ndx):
        """{t: x(t)} of time block ndx (what ``interface.pyomo_model(ndx).x[t].value`` is in the reference)."""
        x_times, _ = self.time_points[ndx]
        v = np.asarray(self.get_primals().get_block(ndx))
        return {t: float(v[i]) for i, t in enumerate(x_times)}

    def p(self, ndx):
        x_times, p_times = self.time_points[ndx]
        v = np.asarray(self.get_primals().get_block(ndx))
        return {t: float(v[len(x_times) + i]) for i, t in enumerate(p_times)}


def _solver_from_class(subproblem_solver_class, subproblem_solver_options, blocks, comm=None):
    """The reference's examples build their linear solver from a sub-solver class and its options
    (``MPISchurComplementLinearSolver(subproblem_solvers={ndx: cls(**opts)}, schur_complement_solver=cls(**opts))``,
    e.g. examples/dynamics.py:166-168); the same call builds this package's class of that name."""
    from parapint_amd.linalg import MPISchurComplementLinearSolver
    from parapint_amd.linalg.comm import SerialComm
    opts = subproblem_solver_options or {}
    return MPISchurComplementLinearSolver(subproblem_solvers={ndx: subproblem_solver_class(**opts) for ndx in blocks},
                                          schur_complement_solver=subproblem_solver_class(**opts),
                                          comm=SerialComm() if comm is None else comm)


def main(linear_solver=None, comm=None, subproblem_solver_class=None, subproblem_solver_options=None, show_plot=False, **problem):
    """dynamics.py:153-180 (without the plot): the default problem -- 90 finite elements, 3 time blocks, the control
    constant over 10 -- through ``ip_solve``; returns the interface.  Either a ready linear solver or, as the reference's
    signature has it, ``subproblem_solver_class`` + ``subproblem_solver_options``."""
    interface = Problem(comm=comm, **problem)
    if linear_solver is None:
        linear_solver = _solver_from_class(subproblem_solver_class, subproblem_solver_options,
                                           interface.local_block_indices, comm)
    options = IPOptions()
    options.linalg.solver = linear_solver
    status = ip_solve(interface=interface, options=options)
    assert status == InteriorPointStatus.optimal
    return interface
