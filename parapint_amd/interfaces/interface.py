"""Interior-point interface of ONE subproblem, Pyomo-free.

The reference's ``InteriorPointInterface`` (parapint/interfaces/interface.py:251-679) wraps a PyNumero NLP
(Pyomo + ASL, neither available here) and builds the primal-dual KKT matrix and right-hand side the
Schur-complement solver is handed (``evaluate_primal_dual_kkt_matrix`` :432-494, ``_rhs`` :496-538).  This module
restates that class for problems given explicitly as a quadratic program

    min  1/2 x' H x + c' x + c0     s.t.  A_eq x = b_eq,   ineq_lb <= A_ineq x <= ineq_ub,   lb <= x <= ub

so that the callers of the hot path (``ip_solve``, the stochastic Schur-complement interface) can be exercised
end to end: same method names, same KKT layout (4 x 4 blocks: primals, slacks, equality duals, inequality
duals), same barrier terms, same regularisation hooks.  It is the producer side of SURVEY.md section 8 rows f2 / C1,
not part of the solver.
"""
import numpy as np
import scipy.sparse as sp
from scipy.sparse import coo_matrix

from parapint_amd.sparse.block_containers import BlockMatrix, BlockVector


class QuadraticProgram(object):
    """Data of one subproblem (all matrices SciPy sparse or None, vectors array-like)."""

    def __init__(self, c, A_eq=None, b_eq=None, A_ineq=None, ineq_lb=None, ineq_ub=None, lb=None, ub=None, H=None,
                 c0=0.0, x0=None, names=None):
        self.c = np.asarray(c, dtype=np.double).ravel()
        n = self.c.size
        self.n = n
        self.H = coo_matrix((n, n)) if H is None else sp.tril(coo_matrix(H)).tocoo()    # lower triangle, as ASL
        self.A_eq = coo_matrix((0, n)) if A_eq is None else coo_matrix(A_eq)
        self.b_eq = np.zeros(self.A_eq.shape[0]) if b_eq is None else np.asarray(b_eq, dtype=np.double).ravel()
        self.A_ineq = coo_matrix((0, n)) if A_ineq is None else coo_matrix(A_ineq)
        m = self.A_ineq.shape[0]
        self.ineq_lb = np.full(m, -np.inf) if ineq_lb is None else np.asarray(ineq_lb, dtype=np.double).ravel()
        self.ineq_ub = np.full(m, np.inf) if ineq_ub is None else np.asarray(ineq_ub, dtype=np.double).ravel()
        self.lb = np.full(n, -np.inf) if lb is None else np.asarray(lb, dtype=np.double).ravel()
        self.ub = np.full(n, np.inf) if ub is None else np.asarray(ub, dtype=np.double).ravel()
        self.c0 = float(c0)
        self.x0 = np.zeros(n) if x0 is None else np.asarray(x0, dtype=np.double).ravel()   # unset Pyomo values start at 0
        self.names = names


def _relaxed(bound, factor, sign):
    """interface.py:389-419: bounds moved outwards by factor * max(1, |bound|)."""
    if factor == 0:
        return bound
    return bound + sign * factor * np.maximum(1.0, np.abs(bound))


class QuadraticProgramNLP(object):
    """A QuadraticProgram behind the NLP protocol the reference's interface talks to (PyNumero's ``ExtendedNLP``, of which
    ``PyomoNLP`` / ``AmplNLP`` are the implementations parapint/interfaces/interface.py:253-256 instantiates): sizes,
    bounds, initial point, the primal / dual state, and the evaluations at that state."""

    def __init__(self, qp):
        self.qp = qp
        self._obj_factor = 1.0
        self._primals = qp.x0.copy()
        self._duals_eq = np.zeros(qp.A_eq.shape[0])
        self._duals_ineq = np.zeros(qp.A_ineq.shape[0])
        self._Hfull = (qp.H + sp.tril(qp.H, -1).T).tocsr()

    def n_primals(self):
        return self.qp.n

    def n_eq_constraints(self):
        return self.qp.A_eq.shape[0]

    def n_ineq_constraints(self):
        return self.qp.A_ineq.shape[0]

    def nnz_hessian_lag(self):
        return self.qp.H.nnz

    def nnz_jacobian_eq(self):
        return self.qp.A_eq.nnz

    def nnz_jacobian_ineq(self):
        return self.qp.A_ineq.nnz

    def primals_lb(self):
        return self.qp.lb

    def primals_ub(self):
        return self.qp.ub

    def ineq_lb(self):
        return self.qp.ineq_lb

    def ineq_ub(self):
        return self.qp.ineq_ub

    def init_primals(self):
        return self.qp.x0

    def init_duals_eq(self):
        return np.zeros(self.qp.A_eq.shape[0])

    def init_duals_ineq(self):
        return np.zeros(self.qp.A_ineq.shape[0])

    def set_primals(self, primals):
        self._primals = np.asarray(primals, dtype=np.double)

    def get_primals(self):
        return self._primals

    def set_duals_eq(self, duals):
        self._duals_eq = np.asarray(duals, dtype=np.double)

    def get_duals_eq(self):
        return self._duals_eq

    def set_duals_ineq(self, duals):
        self._duals_ineq = np.asarray(duals, dtype=np.double)

    def get_duals_ineq(self):
        return self._duals_ineq

    def set_obj_factor(self, obj_factor):
        self._obj_factor = obj_factor

    def get_obj_factor(self):
        return self._obj_factor

    def evaluate_objective(self):
        x = self._primals
        return self._obj_factor * (0.5 * x @ (self._Hfull @ x) + self.qp.c @ x + self.qp.c0)

    def evaluate_grad_objective(self):
        return self._Hfull @ self._primals + self.qp.c

    def evaluate_eq_constraints(self):
        return self.qp.A_eq @ self._primals - self.qp.b_eq

    def evaluate_ineq_constraints(self):
        return self.qp.A_ineq @ self._primals

    def evaluate_jacobian_eq(self):
        return self.qp.A_eq

    def evaluate_jacobian_ineq(self):
        return self.qp.A_ineq

    def evaluate_hessian_lag(self):
        H = self.qp.H
        return coo_matrix((self._obj_factor * H.data, (H.row, H.col)), shape=H.shape)


class CallbackNLP(object):
    """A nonlinear program given by plain functions, behind the NLP protocol (no Pyomo, no ASL):

        min f(x)   s.t.  c_eq(x) = 0,  ineq_lb <= c_ineq(x) <= ineq_ub,  lb <= x <= ub

    grad(x), jac_eq(x), jac_ineq(x) return the gradient and SciPy sparse Jacobians (the same pattern at every x);
    hess_lag(x, y_eq, y_ineq, obj_factor) the Hessian of obj_factor f + y_eq' c_eq + y_ineq' c_ineq (lower triangle or
    full).  Omitted constraint functions mean no such constraints."""

    def __init__(self, x0, f, grad, hess_lag, c_eq=None, jac_eq=None, c_ineq=None, jac_ineq=None, ineq_lb=None, ineq_ub=None,
                 lb=None, ub=None):
        self._x0 = np.asarray(x0, dtype=np.double).ravel()
        n = self._x0.size
        self._f, self._grad, self._hess = f, grad, hess_lag
        self._c_eq = c_eq if c_eq is not None else (lambda x: np.zeros(0))
        self._jac_eq = jac_eq if jac_eq is not None else (lambda x: coo_matrix((0, n)))
        self._c_ineq = c_ineq if c_ineq is not None else (lambda x: np.zeros(0))
        self._jac_ineq = jac_ineq if jac_ineq is not None else (lambda x: coo_matrix((0, n)))
        self._primals = self._x0.copy()
        me, mi = np.asarray(self._c_eq(self._x0)).size, np.asarray(self._c_ineq(self._x0)).size
        self._duals_eq, self._duals_ineq = np.zeros(me), np.zeros(mi)
        self._lb = np.full(n, -np.inf) if lb is None else np.asarray(lb, dtype=np.double).ravel()
        self._ub = np.full(n, np.inf) if ub is None else np.asarray(ub, dtype=np.double).ravel()
        self._ineq_lb = np.full(mi, -np.inf) if ineq_lb is None else np.asarray(ineq_lb, dtype=np.double).ravel()
        self._ineq_ub = np.full(mi, np.inf) if ineq_ub is None else np.asarray(ineq_ub, dtype=np.double).ravel()
        self._obj_factor = 1.0

    def n_primals(self):
        return self._x0.size

    def n_eq_constraints(self):
        return self._duals_eq.size

    def n_ineq_constraints(self):
        return self._duals_ineq.size

    def nnz_hessian_lag(self):
        return coo_matrix(self.evaluate_hessian_lag()).nnz

    def nnz_jacobian_eq(self):
        return coo_matrix(self.evaluate_jacobian_eq()).nnz

    def nnz_jacobian_ineq(self):
        return coo_matrix(self.evaluate_jacobian_ineq()).nnz

    def primals_lb(self):
        return self._lb

    def primals_ub(self):
        return self._ub

    def ineq_lb(self):
        return self._ineq_lb

    def ineq_ub(self):
        return self._ineq_ub

    def init_primals(self):
        return self._x0

    def init_duals_eq(self):
        return np.zeros(self._duals_eq.size)

    def init_duals_ineq(self):
        return np.zeros(self._duals_ineq.size)

    def set_primals(self, primals):
        self._primals = np.asarray(primals, dtype=np.double)

    def get_primals(self):
        return self._primals

    def set_duals_eq(self, duals):
        self._duals_eq = np.asarray(duals, dtype=np.double)

    def get_duals_eq(self):
        return self._duals_eq

    def set_duals_ineq(self, duals):
        self._duals_ineq = np.asarray(duals, dtype=np.double)

    def get_duals_ineq(self):
        return self._duals_ineq

    def set_obj_factor(self, obj_factor):
        self._obj_factor = obj_factor

    def get_obj_factor(self):
        return self._obj_factor

    def evaluate_objective(self):
        return float(self._f(self._primals))

    def evaluate_grad_objective(self):
        return np.asarray(self._grad(self._primals), dtype=np.double)

    def evaluate_eq_constraints(self):
        return np.asarray(self._c_eq(self._primals), dtype=np.double)

    def evaluate_ineq_constraints(self):
        return np.asarray(self._c_ineq(self._primals), dtype=np.double)

    def evaluate_jacobian_eq(self):
        return coo_matrix(self._jac_eq(self._primals))

    def evaluate_jacobian_ineq(self):
        return coo_matrix(self._jac_ineq(self._primals))

    def evaluate_hessian_lag(self):
        return coo_matrix(self._hess(self._primals, self._duals_eq, self._duals_ineq, self._obj_factor))


class InteriorPointInterface(object):
    """Counterpart of ``InteriorPointInterface`` (interface.py:251-679) over ANY object with the NLP protocol it uses
    there (``self._nlp``: PyNumero's ExtendedNLP -- a ``PyomoNLP`` / ``AmplNLP`` where Pyomo is installed, a
    ``QuadraticProgramNLP``, or a hand-written class such as parapint_amd/examples/burgers.py): nonlinear problems
    re-evaluate Hessian and Jacobians at every iterate, exactly as the reference does (:432-494).

    The Hessian of the Lagrangian may come as its lower triangle (ASL's convention, kept by QuadraticProgram) or in
    full; the linear solvers of this package read the lower triangle."""

    def __init__(self, nlp):
        self._nlp = nlp
        self.bounds_relaxation_factor = 0
        self._slacks = self.init_slacks()
        n, mi = nlp.n_primals(), nlp.n_ineq_constraints()
        # interface.py:263-283: ones unless ipopt suffixes exist, zero where the bound is infinite; slack duals from
        # the initial inequality duals according to their sign
        self._init_duals_primals_lb = np.ones(n)
        self._init_duals_primals_ub = np.ones(n)
        self._init_duals_primals_lb[np.isneginf(np.asarray(nlp.primals_lb()))] = 0
        self._init_duals_primals_ub[np.isinf(np.asarray(nlp.primals_ub()))] = 0
        self._duals_primals_lb = self._init_duals_primals_lb.copy()
        self._duals_primals_ub = self._init_duals_primals_ub.copy()
        yi = np.asarray(nlp.init_duals_ineq(), dtype=np.double).reshape(mi)
        self._init_duals_slacks_lb = np.where(yi < 0, 0.0, yi)
        self._init_duals_slacks_ub = np.where(yi > 0, 0.0, yi) * -1.0 + 0.0
        self._duals_slacks_lb = self._init_duals_slacks_lb.copy()
        self._duals_slacks_ub = self._init_duals_slacks_ub.copy()
        self._delta_primals = self._delta_slacks = self._delta_duals_eq = self._delta_duals_ineq = None
        self._barrier = None

    # the state lives in the NLP object, as in the reference
    @property
    def _primals(self):
        return self._nlp.get_primals()

    @property
    def _duals_eq(self):
        return self._nlp.get_duals_eq()

    @property
    def _duals_ineq(self):
        return self._nlp.get_duals_ineq()

    @property
    def _obj_factor(self):
        return self._nlp.get_obj_factor()

    # ---- sizes / options
    def get_bounds_relaxation_factor(self):
        return self.bounds_relaxation_factor

    def set_bounds_relaxation_factor(self, val):
        self.bounds_relaxation_factor = val

    def n_primals(self):
        return self._nlp.n_primals()

    def n_eq_constraints(self):
        return self._nlp.n_eq_constraints()

    def n_ineq_constraints(self):
        return self._nlp.n_ineq_constraints()

    def nnz_hessian_lag(self):
        return self._nlp.nnz_hessian_lag()

    def nnz_jacobian_eq(self):
        return self._nlp.nnz_jacobian_eq()

    def nnz_jacobian_ineq(self):
        return self._nlp.nnz_jacobian_ineq()

    def set_obj_factor(self, obj_factor):
        self._nlp.set_obj_factor(obj_factor)

    def get_obj_factor(self):
        return self._nlp.get_obj_factor()

    # ---- bounds
    def primals_lb(self):
        return _relaxed(np.asarray(self._nlp.primals_lb()), self.bounds_relaxation_factor, -1.0)

    def primals_ub(self):
        return _relaxed(np.asarray(self._nlp.primals_ub()), self.bounds_relaxation_factor, +1.0)

    def ineq_lb(self):
        return _relaxed(np.asarray(self._nlp.ineq_lb()), self.bounds_relaxation_factor, -1.0)

    def ineq_ub(self):
        return _relaxed(np.asarray(self._nlp.ineq_ub()), self.bounds_relaxation_factor, +1.0)

    # ---- initial point
    def init_primals(self):
        return self._nlp.init_primals()

    def init_slacks(self):
        return self._nlp.evaluate_ineq_constraints()              # (interface.py:324-326: at the NLP's current point)

    def init_duals_eq(self):
        return self._nlp.init_duals_eq()

    def init_duals_ineq(self):
        return self._nlp.init_duals_ineq()

    def init_duals_primals_lb(self):
        return self._init_duals_primals_lb

    def init_duals_primals_ub(self):
        return self._init_duals_primals_ub

    def init_duals_slacks_lb(self):
        return self._init_duals_slacks_lb

    def init_duals_slacks_ub(self):
        return self._init_duals_slacks_ub

    # ---- state
    def set_primals(self, primals):
        self._nlp.set_primals(np.asarray(primals, dtype=np.double))

    def set_slacks(self, slacks):
        self._slacks = np.asarray(slacks, dtype=np.double)

    def set_duals_eq(self, duals):
        self._nlp.set_duals_eq(np.asarray(duals, dtype=np.double))

    def set_duals_ineq(self, duals):
        self._nlp.set_duals_ineq(np.asarray(duals, dtype=np.double))

    def set_duals_primals_lb(self, duals):
        self._duals_primals_lb = np.asarray(duals, dtype=np.double)

    def set_duals_primals_ub(self, duals):
        self._duals_primals_ub = np.asarray(duals, dtype=np.double)

    def set_duals_slacks_lb(self, duals):
        self._duals_slacks_lb = np.asarray(duals, dtype=np.double)

    def set_duals_slacks_ub(self, duals):
        self._duals_slacks_ub = np.asarray(duals, dtype=np.double)

    def get_primals(self):
        return self._primals

    def get_slacks(self):
        return self._slacks

    def get_duals_eq(self):
        return self._duals_eq

    def get_duals_ineq(self):
        return self._duals_ineq

    def get_duals_primals_lb(self):
        return self._duals_primals_lb

    def get_duals_primals_ub(self):
        return self._duals_primals_ub

    def get_duals_slacks_lb(self):
        return self._duals_slacks_lb

    def get_duals_slacks_ub(self):
        return self._duals_slacks_ub

    def set_barrier_parameter(self, barrier):
        self._barrier = barrier

    # ---- function evaluations
    def evaluate_objective(self):
        return self._nlp.evaluate_objective()

    def evaluate_grad_objective(self):
        return self._nlp.evaluate_grad_objective()

    def evaluate_eq_constraints(self):
        return self._nlp.evaluate_eq_constraints()

    def evaluate_ineq_constraints(self):
        return self._nlp.evaluate_ineq_constraints()

    def evaluate_jacobian_eq(self):
        return self._nlp.evaluate_jacobian_eq()

    def evaluate_jacobian_ineq(self):
        return self._nlp.evaluate_jacobian_ineq()

    def evaluate_hessian_lag(self):
        return self._nlp.evaluate_hessian_lag()

    def grad_lag_primals_terms(self):
        """J_eq^T y_eq + J_ineq^T y_ineq (the convergence check of the loop needs it, interior_point.py:236-238)."""
        return (self.evaluate_jacobian_eq().T @ self._duals_eq + self.evaluate_jacobian_ineq().T @ self._duals_ineq)

    # ---- the KKT system the linear solver is handed
    def barrier_diagonals(self):
        """The only per-iteration values of a QP's KKT matrix (interface.py:450-465): the primal and the slack
        barrier terms."""
        x, s = self._primals, self._slacks
        dp = self._duals_primals_lb / (x - self.primals_lb()) + self._duals_primals_ub / (self.primals_ub() - x)
        ds = self._duals_slacks_lb / (s - self.ineq_lb()) + self._duals_slacks_ub / (self.ineq_ub() - s)
        return dp, ds

    def evaluate_primal_dual_kkt_matrix(self, timer=None):
        """interface.py:432-494: Hessian + primal barrier diagonal (appended as extra COO entries), slack barrier
        diagonal, both Jacobians with their transposes, -I between slacks and inequality duals, explicit zero
        diagonal blocks for the constraint rows (so that regularisation does not change the pattern)."""
        n, me, mi = self.n_primals(), self.n_eq_constraints(), self.n_ineq_constraints()
        dp, ds = self.barrier_diagonals()
        H = coo_matrix(self.evaluate_hessian_lag())
        low, up = H.row > H.col, H.row < H.col
        if low.any() != up.any():
            # one triangle only (ASL's convention, QuadraticProgram's): the KKT matrix carries both, as the one PyNumero
            # hands the reference's solvers does -- a general-LU sub-solver (ScipyInterface) factorises it as given
            off = low if low.any() else up
            H = coo_matrix((np.concatenate([H.data, H.data[off]]), (np.concatenate([H.row, H.col[off]]),
                                                                     np.concatenate([H.col, H.row[off]]))), shape=H.shape)
        A_eq, A_ineq = coo_matrix(self.evaluate_jacobian_eq()), coo_matrix(self.evaluate_jacobian_ineq())
        idx = np.arange(n)
        hess = coo_matrix((np.concatenate([H.data, dp]), (np.concatenate([H.row, idx]), np.concatenate([H.col, idx]))),
                          shape=(n, n))
        midx = np.arange(mi)
        kkt = BlockMatrix(4, 4)
        kkt.set_block(0, 0, hess)
        kkt.set_block(1, 1, coo_matrix((ds, (midx, midx)), shape=(mi, mi)))
        kkt.set_block(2, 0, A_eq)
        kkt.set_block(0, 2, A_eq.transpose().tocoo())
        kkt.set_block(3, 0, A_ineq)
        kkt.set_block(0, 3, A_ineq.transpose().tocoo())
        neg_eye = coo_matrix((-np.ones(mi), (midx, midx)), shape=(mi, mi))
        kkt.set_block(3, 1, neg_eye)
        kkt.set_block(1, 3, neg_eye.copy())
        eidx = np.arange(me)
        kkt.set_block(2, 2, coo_matrix((np.zeros(me), (eidx, eidx)), shape=(me, me)))
        kkt.set_block(3, 3, coo_matrix((np.zeros(mi), (midx, midx)), shape=(mi, mi)))
        return kkt

    def evaluate_primal_dual_kkt_rhs(self, timer=None):
        """interface.py:496-538 (negative gradient of the barrier Lagrangian and the constraint residuals)."""
        x, s = self._primals, self._slacks
        A_eq, A_ineq = self.evaluate_jacobian_eq(), self.evaluate_jacobian_ineq()
        grad_lag_primals = (self._obj_factor * self.evaluate_grad_objective() + A_eq.T @ self._duals_eq +
                            A_ineq.T @ self._duals_ineq - self._barrier / (x - self.primals_lb()) +
                            self._barrier / (self.primals_ub() - x))
        grad_lag_slacks = (-self._duals_ineq - self._barrier / (s - self.ineq_lb()) +
                           self._barrier / (self.ineq_ub() - s))
        rhs = BlockVector(4)
        rhs.set_block(0, -grad_lag_primals)
        rhs.set_block(1, -grad_lag_slacks)
        rhs.set_block(2, -self.evaluate_eq_constraints())
        rhs.set_block(3, -(self.evaluate_ineq_constraints() - s))
        return rhs

    def set_primal_dual_kkt_solution(self, sol):
        self._delta_primals = np.asarray(sol.get_block(0))
        self._delta_slacks = np.asarray(sol.get_block(1))
        self._delta_duals_eq = np.asarray(sol.get_block(2))
        self._delta_duals_ineq = np.asarray(sol.get_block(3))

    def get_delta_primals(self):
        return self._delta_primals

    def get_delta_slacks(self):
        return self._delta_slacks

    def get_delta_duals_eq(self):
        return self._delta_duals_eq

    def get_delta_duals_ineq(self):
        return self._delta_duals_ineq

    # interface.py:562-588: bound-dual steps recovered from the primal / slack steps
    def get_delta_duals_primals_lb(self):
        return ((self._barrier - self._duals_primals_lb * self._delta_primals) /
                (self._primals - self.primals_lb())) - self._duals_primals_lb

    def get_delta_duals_primals_ub(self):
        return ((self._barrier + self._duals_primals_ub * self._delta_primals) /
                (self.primals_ub() - self._primals)) - self._duals_primals_ub

    def get_delta_duals_slacks_lb(self):
        return ((self._barrier - self._duals_slacks_lb * self._delta_slacks) /
                (self._slacks - self.ineq_lb())) - self._duals_slacks_lb

    def get_delta_duals_slacks_ub(self):
        return ((self._barrier + self._duals_slacks_ub * self._delta_slacks) /
                (self.ineq_ub() - self._slacks)) - self._duals_slacks_ub

    # ---- inertia correction (interface.py:590-619)
    def regularize_equality_gradient(self, kkt, coef, copy_kkt=True):
        if copy_kkt:
            kkt = kkt.copy()
        me, mi = self.n_eq_constraints(), self.n_ineq_constraints()
        kkt.set_block(2, 2, (coef * sp.identity(me, format='coo')).tocoo())
        kkt.set_block(3, 3, (coef * sp.identity(mi, format='coo')).tocoo())
        return kkt

    def regularize_hessian(self, kkt, coef, copy_kkt=True):
        if copy_kkt:
            kkt = kkt.copy()
        hess = kkt.get_block(0, 0)
        kkt.set_block(0, 0, (hess + coef * sp.identity(self.n_primals(), format='coo')).tocoo())
        return kkt


class QPInteriorPointInterface(InteriorPointInterface):
    """The interface over a QuadraticProgram (the Pyomo-free stand-in for a Pyomo model in the tests, the examples and
    the device-resident producers)."""

    def __init__(self, qp):
        self._qp = qp
        super(QPInteriorPointInterface, self).__init__(QuadraticProgramNLP(qp))
