"""Linear solvers behind parapint's ``LinearSolverInterface`` (exports mirror parapint/linalg/__init__.py:1-7; the
MA27 / MUMPS / SciPy sub-solver wrappers are replaced by the batched HIP factorisation, ``HipLDLInterface`` keeps
their single-matrix contract)."""
from .base_linear_solver_interface import LinearSolverInterface
from .results import LinearSolverResults, LinearSolverStatus
from .hip_schur_complement import HipSchurComplementLinearSolver, HipSerialSchurComplementLinearSolver
from .sub_solvers import HipLDLInterface, MumpsInterface, ScipyInterface

# the reference's names for the two classes this package replaces
SchurComplementLinearSolver = HipSerialSchurComplementLinearSolver
MPISchurComplementLinearSolver = HipSchurComplementLinearSolver
InteriorPointMA27Interface = HipLDLInterface
