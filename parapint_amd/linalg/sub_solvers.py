"""Single-matrix sub-solver interfaces by the reference's names, over the same HIP kernels:
``HipLDLInterface`` (= parapint.linalg.InteriorPointMA27Interface, ma27_interface.py:9-256), ``MumpsInterface``
(mumps_interface.py:11-229) and ``ScipyInterface`` (scipy_interface.py:11-67; unsymmetric matrices through general_blocks.py).
Each is a one-block, zero-coupling instance of the batched solver."""
import numpy as np

from parapint_amd.linalg.base_linear_solver_interface import LinearSolverInterface
from parapint_amd.linalg.comm import SerialComm
from parapint_amd.linalg.results import LinearSolverResults, LinearSolverStatus
from parapint_amd.linalg._solver_support import _flat
from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver, _OK


class HipLDLInterface(LinearSolverInterface):
    """Single-matrix sub-solver with the MA27 wrapper's semantics
    (parapint/linalg/ma27_interface.py:9-256): tril is authoritative, inertia is
    (n - neg, neg, 0) on success, singular matrices come back as LinearSolverStatus.singular.
    Implemented as a one-block, zero-coupling instance of the batched solver."""

    @classmethod
    def getLoggerName(cls):
        return 'hip_ldl'

    def __init__(self, cntl_options=None, icntl_options=None, iw_factor=1.2, a_factor=2, engine=None):
        """Same keywords as the reference wrapper (ma27_interface.py:36).  ``cntl_options[1]`` -- MA27's pivot tolerance
        u -- becomes the run-time growth bound |l_ij| <= 1/u of every factorisation and (if larger than the default
        0.01) the threshold of the static pivot choice; the other MA27 controls and the workspace factors have no
        counterpart (storage is sized exactly by the symbolic phase) and are accepted and recorded only."""
        from parapint_amd.sparse.block_containers import BlockMatrix
        self._BlockMatrix = BlockMatrix
        self.cntl_options = dict(cntl_options or {})
        self.icntl_options = dict(icntl_options or {})
        self.iw_factor, self.a_factor = iw_factor, a_factor
        self._engine_arg = engine
        self._sc_made = None
        self._dim = None
        self._num_status = None

    @property
    def _sc(self):
        """The one-block solver behind this interface, created at first use: the reference's callers build one sub-solver
        object per block (``{ndx: InteriorPointMA27Interface(...) for ndx in ...}``) and hand them to the Schur-complement
        solver, which here factorises all blocks as one batch and never calls them -- such placeholders must not each
        open a device handle."""
        if self._sc_made is None:
            u = self.cntl_options.get(1)
            self._sc_made = HipSchurComplementLinearSolver(
                comm=SerialComm(), engine=self._engine_arg, pivot_tolerance=u,
                symbolic_pivot_threshold=None if u is None else max(min(u, 0.5), 0.01),
                general_blocks=getattr(self, 'general_lu', False))
        return self._sc_made

    def _wrap(self, matrix):
        from scipy.sparse import coo_matrix
        n = matrix.shape[0]
        bm = self._BlockMatrix(2, 2)
        bm.set_block(0, 0, matrix)
        bm.set_block(1, 0, coo_matrix((0, n)))
        bm.set_block(1, 1, coo_matrix((0, 0)))
        return bm

    def do_symbolic_factorization(self, matrix, raise_on_error=True, timer=None):
        self._num_status = None
        nrows, ncols = matrix.shape
        if nrows != ncols:
            raise ValueError('Matrix must be square')
        self._dim = nrows
        return self._sc.do_symbolic_factorization(self._wrap(matrix), raise_on_error=raise_on_error, timer=timer)

    def do_numeric_factorization(self, matrix, raise_on_error=True, timer=None):
        if self._dim is None:
            raise RuntimeError('Perform symbolic factorization first!')
        nrows, ncols = matrix.shape
        if nrows != ncols:
            raise ValueError('Matrix must be square')
        if nrows != self._dim:
            raise ValueError('Matrix dimensions do not match the dimensions of '
                             'the matrix used for symbolic factorization')
        res = self._sc.do_numeric_factorization(self._wrap(matrix), raise_on_error=raise_on_error, timer=timer)
        self._num_status = res.status
        return res

    def do_back_solve(self, rhs):
        from parapint_amd.sparse.block_containers import BlockVector
        flat = _flat(rhs)
        bv = BlockVector(2)
        bv.set_block(0, flat)
        bv.set_block(1, np.zeros(0))
        x = self._sc.do_back_solve(bv).get_block(0)
        if hasattr(rhs, 'get_block'):
            out = rhs.copy_structure()
            out.copyfrom(x)
            return out
        return x

    def get_inertia(self):
        if self._num_status is None:
            raise RuntimeError('Must call do_numeric_factorization before inertia can be computed')
        if self._num_status != LinearSolverStatus.successful:
            raise RuntimeError('Can only compute inertia if the numeric factorization was successful.')
        return self._sc.get_inertia()

    def increase_memory_allocation(self, factor):
        self._sc.increase_memory_allocation(factor)


class MumpsInterface(HipLDLInterface):
    """The reference's MUMPS wrapper by name and constructor (parapint/linalg/mumps_interface.py:11-229): ``par``,
    ``comm``, ``cntl_options``, ``icntl_options`` are accepted; CNTL(1) -- MUMPS's relative pivot threshold -- is the run-time
    growth bound, ICNTL(13) / ICNTL(24) are forced as the reference forces them (exact inertia: null pivots are counted,
    not perturbed), the pattern may change between numeric calls (the plan is made again on the union, as
    ``mumps_interface.py:82-83`` re-analyses), and inertia is (n - neg - zero, neg, zero) with the null pivots of
    INFOG(28) (:122-126).  The workspace protocol (ICNTL(23), :105-115) maps to the device value-storage budget."""

    @classmethod
    def getLoggerName(cls):
        return 'mumps'

    def __init__(self, par=1, comm=None, cntl_options=None, icntl_options=None, engine=None, memory_budget_bytes=None):
        icntl = dict(icntl_options or {})
        icntl.setdefault(13, 1)
        icntl.setdefault(24, 0)
        HipLDLInterface.__init__(self, cntl_options=cntl_options, icntl_options=icntl, engine=engine)
        self.par, self.mumps_comm = par, comm
        self._prev_allocation = 0
        self._budget_given = memory_budget_bytes is not None
        if memory_budget_bytes is not None:
            self._sc._eng.set_memory_budget(memory_budget_bytes)
            self._prev_allocation = int(memory_budget_bytes)

    def set_icntl(self, key, value):
        if key == 13 and value <= 0:
            raise ValueError('ICNTL(13) must be positive for the MumpsInterface.')
        if key == 24 and value != 0:
            raise ValueError('ICNTL(24) must be 0 for the MumpsInterface.')
        self.icntl_options[key] = value

    def set_cntl(self, key, value):
        self.cntl_options[key] = value

    def get_icntl(self, key):
        return self.icntl_options.get(key, 0)

    def get_cntl(self, key):
        return self.cntl_options.get(key, 0.0)

    def get_infog(self, key):
        """INFOG(12): negative pivots, INFOG(28): null pivots, INFOG(16) / (18): value storage the plan needs / holds, MB."""
        if key in (12, 28):
            pos, neg, zero = self._sc._inertia if self._sc._inertia is not None else (0, 0, 0)
            return neg if key == 12 else zero
        if key in (16, 18):
            need, have, _ = self._sc._eng.memory_info() if hasattr(self._sc._eng, 'memory_info') else (0, 0, 0)
            return int(round((need if key == 16 else have) / 1e6))
        raise KeyError('INFOG(%d) has no counterpart' % key)

    get_info = get_infog

    def get_inertia(self):
        if self._num_status is None:
            raise RuntimeError('Must call do_numeric_factorization before inertia can be computed')
        if self._sc._inertia is None:
            raise RuntimeError('Can only compute inertia if the numeric factorization was successful.')
        return tuple(int(v) for v in self._sc._inertia)          # null pivots are reported, as INFOG(28) is

    def do_symbolic_factorization(self, matrix, raise_on_error=True, timer=None):
        res = HipLDLInterface.do_symbolic_factorization(self, matrix, raise_on_error=raise_on_error, timer=timer)
        if not self._budget_given:
            self._prev_allocation = self.get_infog(16)          # MB the plan needs (mumps_interface.py:60)
        return res

    def increase_memory_allocation(self, factor):
        """mumps_interface.py:105-115: the new allocation (ICNTL(23), MB; bytes if the budget was given in bytes) is
        factor x the previous one (1 if that rounded to zero) and is returned."""
        self._sc.increase_memory_allocation(factor)
        new_allocation = 1 if self._prev_allocation == 0 else factor * self._prev_allocation
        if not self._budget_given:
            self.icntl_options[23] = new_allocation
        self._prev_allocation = new_allocation
        return new_allocation


class ScipyInterface(HipLDLInterface):
    """The reference's SciPy wrapper by name and constructor (parapint/linalg/scipy_interface.py:11-67), with its
    general-LU semantics for the SOLVE (both triangles read, any square matrix accepted: quirk Q5): a matrix that is not
    exactly symmetric is factorised through its symmetric embedding (general_blocks.py), a symmetric one as the MA27 /
    MUMPS wrappers do.  ``compute_inertia`` keeps its meaning -- without it ``get_inertia`` raises (:64-67) --, but the
    inertia of an unsymmetric matrix (the reference: eigenvalues counted by a dense eigensolver on the host, :39-44) is
    not offered: ``get_inertia`` raises after such a factorisation.  Handing objects of this class to the Schur-complement
    solver as its ``subproblem_solvers`` selects the same semantics for the blocks of a block-bordered matrix."""

    general_lu = True

    @classmethod
    def getLoggerName(cls):
        return 'scipy'

    def __init__(self, compute_inertia=False, engine=None):
        HipLDLInterface.__init__(self, engine=engine)
        self.compute_inertia = compute_inertia

    def get_inertia(self):
        if not self.compute_inertia:
            raise RuntimeError('The intertia was not computed during factorization. Set compute_inertia to True.')
        return HipLDLInterface.get_inertia(self)
