from .interface import QPInteriorPointInterface, QuadraticProgram
from .schur_complement.sc_ip_interface import (MPIStochasticSchurComplementInteriorPointInterface,
                                               StochasticSchurComplementInteriorPointInterface)
