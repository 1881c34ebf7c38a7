"""The known answers of the reference's algorithm tests (parapint/algorithms/tests/test_interior_point.py:22-213 and
test_reg.py:17-125), re-expressed for the restated loop and the single-problem ``InteriorPointInterface`` over NLP objects
written out by hand (the reference states them in Pyomo): optimal points and multipliers of two small NLPs, the
inertia-correction loop on a matrix that needs it, a concave problem that needs regularisation along the way, and the
helper functions' tables.  Linear solvers: the oracle's ScipyInterface(compute_inertia=True) as in the reference's tests,
and the product's single-matrix LDL^T interface (its MA27 stand-in) on the numpy engine and on the device."""
import math

import numpy as np
import pytest
from scipy.sparse import coo_matrix

from parapint_amd.algorithms.interior_point import (IPOptions, InteriorPointStatus, _frac_lb, _frac_ub, ip_solve,
                                                    numeric_factorization, process_init, process_init_duals_lb,
                                                    process_init_duals_ub)
from parapint_amd.interfaces.interface import CallbackNLP, InteriorPointInterface


def _dense(rows):
    return coo_matrix(np.asarray(rows, dtype=np.double))


def problem_1():
    """min x^2 + y^2  s.t.  y == exp(x),  y >= (x - 1)^2   (test_interior_point.py:23-29; the inequality in the normal
    form Pyomo writes it: (x - 1)^2 - y <= 0)."""
    return CallbackNLP(
        x0=[0.0, 0.0], f=lambda v: v[0] ** 2 + v[1] ** 2, grad=lambda v: 2.0 * v,
        hess_lag=lambda v, ye, yi, of: _dense([[2.0 * of - ye[0] * math.exp(v[0]) + 2.0 * yi[0], 0.0], [0.0, 2.0 * of]]),
        c_eq=lambda v: np.array([v[1] - math.exp(v[0])]), jac_eq=lambda v: _dense([[-math.exp(v[0]), 1.0]]),
        c_ineq=lambda v: np.array([(v[0] - 1.0) ** 2 - v[1]]), jac_ineq=lambda v: _dense([[2.0 * (v[0] - 1.0), -1.0]]),
        ineq_ub=[0.0])


def problem_2():
    """min x^2,  1 <= x <= 4   (:47-56)."""
    return CallbackNLP(x0=[0.0], f=lambda v: v[0] ** 2, grad=lambda v: 2.0 * v, hess_lag=lambda v, ye, yi, of: _dense([[2.0 * of]]),
                       lb=[1.0], ub=[4.0])


def regularization_model():
    """test_reg.py:17-34: min F^2 s.t. 1 == x1 + x2 + x3, F x_i == f_i with f_1 = 1, f_2 = 2 fixed, f_3 free; all
    variables start at 0.  Variables here: x1, x2, x3, f3, F (fixed variables are constants, as the NL writer makes them)."""
    fixed = (1.0, 2.0)

    def c_eq(v):
        return np.array([v[0] + v[1] + v[2] - 1.0, v[4] * v[0] - fixed[0], v[4] * v[1] - fixed[1], v[4] * v[2] - v[3]])

    def jac(v):
        F = v[4]
        return _dense([[1, 1, 1, 0, 0], [F, 0, 0, 0, v[0]], [0, F, 0, 0, v[1]], [0, 0, F, -1, v[2]]])

    def hess(v, ye, yi, of):
        H = np.zeros((5, 5))
        H[4, 4] = 2.0 * of
        H[4, 0], H[4, 1], H[4, 2] = ye[1], ye[2], ye[3]          # d2 (F x_i) / dF dx_i (lower triangle)
        return _dense(H)
    return CallbackNLP(x0=np.zeros(5), f=lambda v: v[4] ** 2, grad=lambda v: np.array([0, 0, 0, 0, 2.0 * v[4]]),
                       hess_lag=hess, c_eq=c_eq, jac_eq=jac)


def regularization_model_2():
    """test_reg.py:37-43: min -x^2 - y^2 s.t. y <= exp(-x), 0 <= x, y <= 1, start (0.1, 0.1): optimum (1, exp(-1))."""
    return CallbackNLP(
        x0=[0.1, 0.1], f=lambda v: -v[0] ** 2 - v[1] ** 2, grad=lambda v: -2.0 * v,
        hess_lag=lambda v, ye, yi, of: _dense([[-2.0 * of - yi[0] * math.exp(-v[0]), 0.0], [0.0, -2.0 * of]]),
        c_ineq=lambda v: np.array([v[1] - math.exp(-v[0])]), jac_ineq=lambda v: _dense([[math.exp(-v[0]), 1.0]]),
        ineq_ub=[0.0], lb=[0.0, 0.0], ub=[1.0, 1.0])


def _scipy_solver():
    from oracle.subsolvers import ScipyInterface
    return ScipyInterface(compute_inertia=True)


def _product_ldl_on_cpu():
    from hostsim_engine import HostSimEngine
    from parapint_amd.linalg.hip_schur_complement import HipLDLInterface
    return HipLDLInterface(engine=HostSimEngine())


def _product_ldl_on_device():
    from parapint_amd.linalg.hip_schur_complement import HipLDLInterface
    return HipLDLInterface()


def _known_answers(make_solver):
    it = InteriorPointInterface(problem_1())
    opt = IPOptions()
    opt.linalg.solver = make_solver()
    assert ip_solve(interface=it, options=opt) == InteriorPointStatus.optimal
    x = it.get_primals()
    assert round(x[0] - 0, 7) == 0 and round(x[1] - 1, 7) == 0                                # :37-38
    assert round(it.get_duals_eq()[0] - (-1 - 1.0 / 3.0), 7) == 0                             # :39
    assert round(it.get_duals_ineq()[0] - 2.0 / 3.0, 7) == 0                                  # :40
    it = InteriorPointInterface(problem_2())
    opt = IPOptions()
    opt.linalg.solver = make_solver()
    assert ip_solve(interface=it, options=opt) == InteriorPointStatus.optimal
    assert round(it.get_primals()[0] - 1, 7) == 0                                             # :56
    # test_reg.py:46-71: at the initial point the KKT matrix has the wrong inertia; the loop finds a coefficient
    it = InteriorPointInterface(regularization_model())
    opt = IPOptions()
    solver = opt.linalg.solver = make_solver()
    it.set_barrier_parameter(1e-1)
    kkt = it.evaluate_primal_dual_kkt_matrix()
    solver.do_symbolic_factorization(kkt)
    reg_coef = numeric_factorization(interface=it, kkt=kkt, options=opt, inertia_coef=opt.inertia_correction.init_coef)
    assert reg_coef >= 1e-8
    n_pos, n_neg, n_null = solver.get_inertia()
    assert n_null == 0 and n_neg == it.n_eq_constraints() + it.n_ineq_constraints()
    # test_reg.py:93-104
    it = InteriorPointInterface(regularization_model_2())
    opt = IPOptions()
    opt.linalg.solver = make_solver()
    assert ip_solve(interface=it, options=opt) == InteriorPointStatus.optimal
    x = it.get_primals()
    assert round(x[0] - 1, 7) == 0 and round(x[1] - math.exp(-1), 7) == 0


def test_known_answers_over_the_oracle_scipy_interface():
    _known_answers(_scipy_solver)
    # examples/tests/test_examples.py:10-16 (examples/interior_point.py)
    from parapint_amd.examples import interior_point as ex
    x = ex.main(linear_solver=_scipy_solver()).get_primals()
    assert round(x[0] - 0, 7) == 0 and round(x[1] - 1, 7) == 0


def test_known_answers_over_the_product_ldl_interface_on_the_cpu_engine():
    _known_answers(_product_ldl_on_cpu)


@pytest.mark.gpu
def test_known_answers_over_the_product_ldl_interface_on_the_device():
    _known_answers(_product_ldl_on_device)


def test_process_init_tables():
    """test_interior_point.py:102-142."""
    lb, ub = np.array([-np.inf, -np.inf, -2, -2]), np.array([np.inf, 2, np.inf, 2])
    for start, expected in ((0, [0, 0, 0, 0]), (-2, [-2, -2, -1, 0]), (-3, [-3, -3, -1, 0]), (2, [2, 1, 2, 0]), (3, [3, 1, 3, 0])):
        x = np.full(4, float(start))
        process_init(x, lb, ub)
        assert np.allclose(x, expected)
    lb = np.array([-5, 0, -np.inf, 2], dtype=np.double)
    for start in (0.0, -1.0):
        x = np.full(4, start)
        process_init_duals_lb(x, lb)
        assert np.allclose(x, [1, 1, 0, 1])
    x = np.full(4, 2.0)
    process_init_duals_ub(x, np.array([-5, 0, np.inf, 2], dtype=np.double))
    assert np.allclose(x, [2, 2, 0, 2])


def test_fraction_to_the_boundary_tables():
    """test_interior_point.py:145-213."""
    tau, x = 0.9, np.zeros(4)
    xl, xu = np.array([-np.inf, -1, -np.inf, -1]), np.array([np.inf, 1, np.inf, 1])
    table = (([-0.1] * 4, 1), ([-1] * 4, 0.9), ([-10] * 4, 0.09), ([1] * 4, 1), ([-10, 1, -10, 1], 1),
             ([-10, -1, -10, -1], 0.9), ([1, -10, 1, -1], 0.09))
    for dx, alpha in table:
        dx = np.asarray(dx, dtype=np.double)
        assert round(_frac_lb(tau, x, dx, xl) - alpha, 7) == 0
        assert round(_frac_ub(tau, x, -dx, xu) - alpha, 7) == 0         # the upper-bound table is its mirror image
