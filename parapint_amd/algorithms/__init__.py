from .interior_point import (IPOptions, InteriorPointStatus, InertiaCorrectionOptions, LinalgOptions, check_convergence,
                             fraction_to_the_boundary, ip_solve, numeric_factorization,
                             try_factorization_and_reallocation)
