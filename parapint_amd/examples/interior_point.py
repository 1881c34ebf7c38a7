"""parapint/examples/interior_point.py, Pyomo-free:  min x^2 + y^2  s.t.  y >= (x - 1)^2,  y == exp(x)  through
``InteriorPointInterface`` and ``ip_solve`` with a single-matrix linear solver (the reference's test expects x = 0, y = 1,
examples/tests/test_examples.py:10-16).  The model is written out as functions (``CallbackNLP``) where the reference builds
a Pyomo model; the inequality is in the normal form Pyomo gives it, (x - 1)^2 - y <= 0."""
import math

import numpy as np
from scipy.sparse import coo_matrix

from parapint_amd.algorithms.interior_point import IPOptions, InteriorPointStatus, ip_solve
from parapint_amd.interfaces.interface import CallbackNLP, InteriorPointInterface


def build_model():
    return CallbackNLP(
        x0=[0.0, 0.0],
        f=lambda v: v[0] ** 2 + v[1] ** 2,
        grad=lambda v: 2.0 * v,
        hess_lag=lambda v, y_eq, y_ineq, obj_factor: coo_matrix(np.array(
            [[2.0 * obj_factor - y_eq[0] * math.exp(v[0]) + 2.0 * y_ineq[0], 0.0], [0.0, 2.0 * obj_factor]])),
        c_eq=lambda v: np.array([v[1] - math.exp(v[0])]),
        jac_eq=lambda v: coo_matrix(np.array([[-math.exp(v[0]), 1.0]])),
        c_ineq=lambda v: np.array([(v[0] - 1.0) ** 2 - v[1]]),
        jac_ineq=lambda v: coo_matrix(np.array([[2.0 * (v[0] - 1.0), -1.0]])),
        ineq_ub=[0.0])


def main(linear_solver):
    """Returns the interface after a successful solve (``interface.get_primals()`` = [x, y])."""
    interface = InteriorPointInterface(build_model())
    options = IPOptions()
    options.linalg.solver = linear_solver
    status = ip_solve(interface=interface, options=options)
    assert status == InteriorPointStatus.optimal
    return interface


if __name__ == '__main__':
    from parapint_amd.linalg.hip_schur_complement import HipLDLInterface
    print(main(linear_solver=HipLDLInterface(cntl_options={1: 1e-6})).get_primals())
