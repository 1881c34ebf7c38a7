"""The plug-in boundary of the hot path.

Same five methods, keyword names and defaults as the reference ABC
(``parapint/linalg/base_linear_solver_interface.py:5-56``).  ``ip_solve`` calls
them by keyword (``parapint/algorithms/interior_point.py:646``), so the names are
part of the contract.
"""
from abc import ABC, abstractmethod
import logging


class LinearSolverInterface(ABC):
    @classmethod
    def getLoggerName(cls):
        return 'linear_solver'

    @classmethod
    def getLogger(cls):
        name = 'algorithms.' + cls.getLoggerName()
        return logging.getLogger(name)

    @abstractmethod
    def do_symbolic_factorization(self, matrix, raise_on_error=True, timer=None):
        pass

    @abstractmethod
    def do_numeric_factorization(self, matrix, raise_on_error=True, timer=None):
        pass

    def increase_memory_allocation(self, factor):
        raise NotImplementedError('Should be implemented by base class.')

    @abstractmethod
    def do_back_solve(self, rhs):
        pass

    @abstractmethod
    def get_inertia(self):
        pass
