"""The vector work of the interior-point step on device-resident vectors.

After ``do_back_solve`` has left the step in HBM (``DeviceBlockVector``), what ``ip_solve`` does with it is
elementwise work and a handful of reductions (parapint/algorithms/interior_point.py:174-317 convergence check,
:655-758 fraction to the boundary, :619-626 the step itself; on PyNumero's MPIBlockVector each of them hides a
scalar all-reduce).  These functions run them as fused HIP kernels on the solver's stream (include/parapint_hip.h:
pp_vec_step_stats / pp_vec_max_abs / pp_vec_axpy); arguments are torch tensors (float64, contiguous, on the solver's
device) of equal length, any shape.  Across ranks the four scalars are combined by the caller with one MIN / MAX
all-reduce, as the reference's MPIBlockVector does per reduction.
"""
import ctypes

import numpy as np


def _ptr(t, n):
    if t is None:
        return None
    if t.numel() != n or not t.is_contiguous() or not t.is_cuda or str(t.dtype) != 'torch.float64':
        raise ValueError('expected contiguous float64 device tensors of %d elements' % n)
    return t.data_ptr()


def step_stats(solver, x, dx=None, lb=None, ub=None, zl=None, dzl=None, zu=None, dzu=None, tau=1.0, barrier=0.0):
    """(alpha_primal_max, alpha_dual_max, complementarity residual of the lower bounds, of the upper bounds) of one
    variable family in one pass (interior_point.py:655-758 with tau, :257-266 with the barrier parameter)."""
    eng = solver._eng
    n = x.numel()
    out = np.zeros(4)
    eng.ns.check(eng.lib.pp_vec_step_stats(eng.ns.h, ctypes.c_int64(n), _ptr(x, n), _ptr(dx, n), _ptr(lb, n), _ptr(ub, n),
                                           _ptr(zl, n), _ptr(dzl, n), _ptr(zu, n), _ptr(dzu, n), float(tau), float(barrier),
                                           out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))), 'pp_vec_step_stats')
    return float(out[0]), float(out[1]), float(out[2]), float(out[3])


def max_abs(solver, v):
    eng = solver._eng
    out = ctypes.c_double(0.0)
    eng.ns.check(eng.lib.pp_vec_max_abs(eng.ns.h, ctypes.c_int64(v.numel()), _ptr(v, v.numel()), ctypes.byref(out)),
                 'pp_vec_max_abs')
    return float(out.value)


def axpy_(solver, y, alpha, x):
    """y += alpha * x in place."""
    eng = solver._eng
    n = y.numel()
    eng.ns.check(eng.lib.pp_vec_axpy(eng.ns.h, ctypes.c_int64(n), float(alpha), _ptr(x, n), _ptr(y, n)), 'pp_vec_axpy')
    return y
