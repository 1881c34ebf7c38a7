"""MI355X-native Schur-complement KKT solver behind parapint's ``LinearSolverInterface`` and the callers around it; the
sub-packages mirror the reference's (parapint/__init__.py:1-3): ``linalg``, ``interfaces``, ``algorithms``."""
