"""Oracle restatement of the reference's sub-block solvers.  TEST INFRASTRUCTURE ONLY.

``ScipyInterface``  follows parapint/linalg/scipy_interface.py:11-67 (general SuperLU
of the matrix *as given*, optional dense-eigenvalue inertia with a 1e-8 cut).
``SymmetricLDLInterface`` follows the MA27 wrapper's *semantics*
(parapint/linalg/ma27_interface.py:52-203: lower triangle is authoritative, inertia
reported as (n - neg, neg, 0)); HSL MA27 itself is not available, so the
factorisation is LAPACK's Bunch-Kaufman ``scipy.linalg.ldl`` on the dense
symmetrised matrix -- small fixtures only.
"""
import numpy as np
from scipy.linalg import eigvals, ldl
from scipy.sparse import isspmatrix_csc, tril
from scipy.sparse.linalg import splu

from parapint_amd.linalg.base_linear_solver_interface import LinearSolverInterface
from parapint_amd.linalg.results import LinearSolverResults, LinearSolverStatus


def _is_block_vector(v):
    return hasattr(v, 'get_block') and hasattr(v, 'copy_structure')


class ScipyInterface(LinearSolverInterface):
    def __init__(self, compute_inertia=False):
        self._lu = None
        self._inertia = None
        self.compute_inertia = compute_inertia

    def do_symbolic_factorization(self, matrix, raise_on_error=True, timer=None):
        # scipy_interface.py:20-23: symbolic is a no-op
        return LinearSolverResults(LinearSolverStatus.successful)

    def do_numeric_factorization(self, matrix, raise_on_error=True, timer=None):
        # scipy_interface.py:25-47
        if not isspmatrix_csc(matrix):
            matrix = matrix.tocsc()
        res = LinearSolverResults()
        try:
            self._lu = splu(matrix)
            res.status = LinearSolverStatus.successful
        except RuntimeError as err:
            if raise_on_error:
                raise err
            if 'Factor is exactly singular' in str(err):
                res.status = LinearSolverStatus.singular
            else:
                res.status = LinearSolverStatus.error
        if self.compute_inertia:
            eig = eigvals(matrix.toarray())
            pos = int(np.count_nonzero(eig > 1e-8))
            neg = int(np.count_nonzero(eig < -1e-8))
            self._inertia = (pos, neg, len(eig) - pos - neg)
        return res

    def do_back_solve(self, rhs):
        # scipy_interface.py:49-62: ndarray or BlockVector in, same kind out
        _rhs = rhs.flatten() if _is_block_vector(rhs) else rhs
        result = self._lu.solve(_rhs)
        if _is_block_vector(rhs):
            out = rhs.copy_structure()
            out.copyfrom(result)
            result = out
        return result

    def get_inertia(self):
        if self._inertia is None:
            raise RuntimeError('The intertia was not computed during do_numeric_factorization. '
                               'Set compute_inertia to True.')
        return self._inertia

    def increase_memory_allocation(self, factor):
        pass


class SymmetricLDLInterface(LinearSolverInterface):
    """MA27-wrapper semantics on a dense Bunch-Kaufman LDL^T (small matrices)."""

    def __init__(self):
        self._dim = None
        self._num_status = None
        self._lu = None
        self._neg = None

    def do_symbolic_factorization(self, matrix, raise_on_error=True, timer=None):
        # ma27_interface.py:67-92: tril, square check, pattern only
        self._num_status = None
        m = tril(matrix.tocoo())
        if m.shape[0] != m.shape[1]:
            raise ValueError('Matrix must be square')
        self._dim = m.shape[0]
        return LinearSolverResults(LinearSolverStatus.successful)

    def do_numeric_factorization(self, matrix, raise_on_error=True, timer=None):
        # ma27_interface.py:110-140
        if self._dim is None:
            raise RuntimeError('Perform symbolic factorization first!')
        low = tril(matrix.tocoo()).toarray()
        if low.shape[0] != self._dim:
            raise ValueError('Matrix dimensions do not match the dimensions of '
                             'the matrix used for symbolic factorization')
        full = low + low.T - np.diag(np.diag(low))
        res = LinearSolverResults(LinearSolverStatus.successful)
        _, d, _ = ldl(full, lower=True)
        # eigenvalues of the block-diagonal D give the inertia (Sylvester)
        ev = np.linalg.eigvalsh(d)
        scale = max(1.0, float(np.abs(full).max()))
        if np.any(np.abs(ev) <= 1e-13 * scale):
            if raise_on_error:
                raise RuntimeError('Numeric factorization was not successful; return code: -5')
            res.status = LinearSolverStatus.singular
        self._neg = int(np.count_nonzero(ev < 0))
        self._full = full
        self._num_status = res.status
        return res

    def do_back_solve(self, rhs):
        _rhs = rhs.flatten() if _is_block_vector(rhs) else np.asarray(rhs, dtype=np.double)
        result = np.linalg.solve(self._full, _rhs)
        if _is_block_vector(rhs):
            out = rhs.copy_structure()
            out.copyfrom(result)
            result = out
        return result

    def get_inertia(self):
        # ma27_interface.py:197-203
        if self._num_status is None:
            raise RuntimeError('Must call do_numeric_factorization before inertia can be computed')
        if self._num_status != LinearSolverStatus.successful:
            raise RuntimeError('Can only compute inertia if the numeric factorization was successful.')
        return (self._dim - self._neg, self._neg, 0)

    def increase_memory_allocation(self, factor):
        pass
