"""ctypes driver of the TEST-ONLY host interpreter (tests/hostsim/hostsim.cpp)."""
import ctypes
import os
import subprocess

import numpy as np
from scipy.sparse import coo_matrix, tril

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
_LIB = None


def build_hostsim(force=False):
    so = os.path.join(HERE, 'hostsim', 'libpp_hostsim.so')
    srcs = [os.path.join(HERE, 'hostsim', 'hostsim.cpp'),
            os.path.join(ROOT, 'parapint_amd', 'csrc', 'symbolic.cpp'),
            os.path.join(ROOT, 'parapint_amd', 'csrc', 'plan.hpp'),
            os.path.join(ROOT, 'parapint_amd', 'csrc', 'switches.hpp'),
            os.path.join(ROOT, 'parapint_amd', 'csrc', 'pivot.hpp'),
            os.path.join(ROOT, 'parapint_amd', 'csrc', 'dense_bk.hpp')]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-pthread', '-shared', '-fPIC', '-o', so, srcs[0], srcs[1]])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build_hostsim())
        L.ppsim_create.restype = ctypes.c_void_p
        L.ppsim_error.restype = ctypes.c_char_p
        _LIB = L
    return _LIB


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


class HostSim(object):
    """Plan + one-instance numeric interpreter for a block K (scipy sparse, lower triangle
    authoritative) and border A (n_c x n)."""

    def __init__(self, K, A, pattern_only=False, acc_doubles=0, delta_abs=-1, delta_rel=-1.0):
        L = lib()
        Kl = tril(coo_matrix(K)).tocsc()
        Kl.sum_duplicates()
        Kl = Kl.tocoo()
        self.n = K.shape[0]
        self.rowK = np.ascontiguousarray(Kl.row, dtype=np.int32)
        self.colK = np.ascontiguousarray(Kl.col, dtype=np.int32)
        Ac = coo_matrix(A).tocsr()
        Ac.sum_duplicates()
        Ac = Ac.tocoo()
        self.nc = A.shape[0]
        self.rowB = np.ascontiguousarray(Ac.row, dtype=np.int32)
        self.colB = np.ascontiguousarray(Ac.col, dtype=np.int32)
        self.can0 = np.concatenate([Kl.data, Ac.data]).astype(np.double)
        self.h = ctypes.c_void_p(L.ppsim_create(self.n, self.nc, self.rowK.size, _ip(self.rowK), _ip(self.colK),
                                                self.rowB.size, _ip(self.rowB), _ip(self.colB),
                                                None if pattern_only else _dp(self.can0),
                                                int(acc_doubles), int(delta_abs), ctypes.c_double(delta_rel)))
        err = L.ppsim_error(self.h)
        if err:
            raise RuntimeError(err.decode())
        st = np.zeros(13, dtype=np.int64)
        L.ppsim_stats(self.h, st.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
        keys = ['n', 'nc', 'npiv', 'n_levels', 'n_2x2', 'usize', 'nnz_L', 'flops_factor', 'flops_schur',
                'ntasks', 'nentries', 'ntiles', 'ntilerecs']
        self.stats = dict(zip(keys, [int(v) for v in st]))
        self.stats['tail_level0'] = self.stats['n_2x2'] // 1000000
        self.stats['n_2x2'] = self.stats['n_2x2'] % 1000000

    def canonical(self, K, A):
        Kl = tril(coo_matrix(K)).tocsc()
        Kl.sum_duplicates()
        Kl = Kl.tocoo()
        Ac = coo_matrix(A).tocsr()
        Ac.sum_duplicates()
        Ac = Ac.tocoo()
        assert np.array_equal(Kl.row, self.rowK) and np.array_equal(Kl.col, self.colK)
        return np.concatenate([Kl.data, Ac.data]).astype(np.double)

    def perm(self):
        p = np.zeros(self.n, dtype=np.int32)
        lib().ppsim_get_perm(self.h, _ip(p))
        return p

    def level_task_counts(self):
        c = np.zeros(self.stats['n_levels'], dtype=np.int32)
        lib().ppsim_get_level_task_counts(self.h, _ip(c))
        return c

    def factor(self, can=None, eps=1e-13):
        can = self.can0 if can is None else np.ascontiguousarray(can, dtype=np.double)
        self.U = np.zeros(self.stats['usize'])
        self.L = np.zeros(self.stats['usize'])
        self.Dinv = np.zeros(lib().ppsim_dsize(self.h))
        S = np.zeros((self.nc, self.nc))
        inertia = np.zeros(3, dtype=np.int64)
        rc = lib().ppsim_factor(self.h, _dp(can), _dp(self.U), _dp(self.L), _dp(self.Dinv), _dp(S),
                                inertia.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), ctypes.c_double(eps))
        S = np.tril(S) + np.tril(S, -1).T
        return rc, S, tuple(int(v) for v in inertia)

    def forward(self, rhs):
        W = np.zeros(self.n + self.nc)
        rhs = np.ascontiguousarray(rhs, dtype=np.double)
        lib().ppsim_forward(self.h, _dp(self.L), _dp(rhs), _dp(W))
        return W

    def backward(self, W, xc):
        X = np.zeros(self.n + self.nc)
        X[self.n:] = xc
        x = np.zeros(self.n)
        lib().ppsim_backward(self.h, _dp(self.L), _dp(self.Dinv), _dp(np.ascontiguousarray(W)), _dp(X), _dp(x))
        return x

    def __del__(self):
        try:
            lib().ppsim_destroy(self.h)
        except Exception:
            pass


def bk_factor_solve(S, rhs, eps=1e-14):
    """Dense Bunch-Kaufman LDL^T (the algorithm the GPU runs on S) on the host: returns x, inertia."""
    n = S.shape[0]
    A = np.asfortranarray(np.tril(S), dtype=np.double).copy(order='F')
    ipiv = np.zeros(n, dtype=np.int32)
    info = np.zeros(3, dtype=np.int32)
    lib().ppsim_bk_factor(n, _dp(A), _ip(ipiv), _ip(info), ctypes.c_double(eps))
    b = np.ascontiguousarray(rhs, dtype=np.double).copy()
    lib().ppsim_bk_solve(n, _dp(A), _ip(ipiv), _dp(b))
    return b, tuple(int(v) for v in info)
