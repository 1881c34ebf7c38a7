"""Exports mirror parapint/interfaces/__init__.py:1-3 (the single-problem interface, the stochastic and the dynamic
Schur-complement interfaces with their MPI flavours); plus the Pyomo-free model objects they are given here."""
from .interface import (CallbackNLP, InteriorPointInterface, QPInteriorPointInterface, QuadraticProgram,
                        QuadraticProgramNLP)
from .schur_complement.sc_ip_interface import (DynamicSchurComplementInteriorPointInterface,
                                               MPIDynamicSchurComplementInteriorPointInterface,
                                               MPIStochasticSchurComplementInteriorPointInterface,
                                               StochasticSchurComplementInteriorPointInterface)
