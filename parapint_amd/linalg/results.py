"""Status types of the linear-solver boundary.

Mirrors the reference's ``parapint/linalg/results.py:4-14`` (same enum names and
integer values; the integer values are also the status codes returned across the
C-ABI declared in ``include/parapint_hip.h``).
"""
import enum


class LinearSolverStatus(enum.Enum):
    successful = 0
    not_enough_memory = 1
    singular = 2
    error = 3
    warning = 4


class LinearSolverResults(object):
    def __init__(self, status=None):
        self.status = status
