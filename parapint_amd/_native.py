"""ctypes binding of libparapint_hip.so (the C ABI declared in include/parapint_hip.h).

There is no CPU fallback: if the library is missing or no HIP device is usable the
product raises.  Build with ``python -c 'import __graft_entry__ as g; g.build()'``.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libparapint_hip.so')
if os.environ.get('PP_LIB_VARIANT'):      # kernel experiments: an alternative build of the same sources, next to the product library
    LIB_PATH = os.path.join(_HERE, 'csrc', 'libparapint_hip_%s.so' % os.environ['PP_LIB_VARIANT'])

_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_f64p = ctypes.POINTER(ctypes.c_double)

# name -> (restype, argtypes); every symbol include/parapint_hip.h declares
SIGNATURES = {
    'pp_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p]),
    'pp_destroy': (None, [ctypes.c_void_p]),
    'pp_last_error': (ctypes.c_char_p, [ctypes.c_void_p]),
    'pp_begin_symbolic': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'pp_add_group': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i32p, _i32p,
                                    ctypes.c_int, _i32p, _i32p, ctypes.c_int, _i32p, _i32p, _f64p,
                                    ctypes.POINTER(ctypes.c_int)]),
    'pp_add_group_mapped': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i32p, _i32p,
                                           ctypes.c_int, _i32p, _i32p, ctypes.c_int, _i32p, _i32p, _f64p, ctypes.c_int, _i32p,
                                           ctypes.POINTER(ctypes.c_int)]),
    'pp_set_coupling_structure': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    'pp_set_coupling_schedule': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'pp_schur_buffer_doubles': (ctypes.c_int64, [ctypes.c_void_p]),
    'pp_end_symbolic': (ctypes.c_int, [ctypes.c_void_p]),
    'pp_upload_values': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]),
    'pp_upload_values_compact': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int]),
    'pp_used_raw_entries': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _i32p, ctypes.c_int]),
    'pp_set_value_map': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _i32p, _f64p]),
    'pp_source_buffer': (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int]),
    'pp_bind_source_buffer': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    'pp_upload_sources': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]),
    'pp_bind_native_vectors': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    'pp_solve_coupling_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    'pp_coupling_solution_buffer': (ctypes.c_void_p, [ctypes.c_void_p]),
    'pp_copy_coupling_solution': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    'pp_bind_solution_buffer': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    'pp_stage_values_runs': (ctypes.c_int, [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 8 +
                             [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                              ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                              ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    'pp_stage_upload_compact': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 8 +
                                [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                 ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                 ctypes.c_void_p, ctypes.c_void_p]),
    'pp_stage_upload_verified_begin': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                      ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
                                                      ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'pp_stage_upload_end': (ctypes.c_int, [ctypes.c_void_p]),
    'pp_upload_rhs_rows': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    'pp_download_solution_rows': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    'pp_copy_rows': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]),
    'pp_host_alloc': (ctypes.c_void_p, [ctypes.c_int64]),
    'pp_host_free': (None, [ctypes.c_void_p]),
    'pp_raw_buffer': (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int]),
    'pp_bind_raw_buffer': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    'pp_numeric_local': (ctypes.c_int, [ctypes.c_void_p]),
    'pp_numeric_factor_blocks': (ctypes.c_int, [ctypes.c_void_p]),
    'pp_numeric_schur': (ctypes.c_int, [ctypes.c_void_p]),
    'pp_numeric_schur_ex': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'pp_fail_local': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'pp_schur_buffer': (ctypes.c_void_p, [ctypes.c_void_p]),
    'pp_bind_schur_buffer': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    'pp_factor_schur': (ctypes.c_int, [ctypes.c_void_p, _f64p]),
    'pp_factor_schur_corner': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, _i64p, _f64p]),
    'pp_set_supernodes': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    'pp_set_instance_splits': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'pp_set_dense_policy': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'pp_get_dense_mode': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]),
    'pp_get_status': (ctypes.c_int, [ctypes.c_void_p, _i64p]),
    'pp_get_schur': (ctypes.c_int, [ctypes.c_void_p, _f64p]),
    'pp_upload_rhs': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]),
    'pp_rhs_buffer': (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int]),
    'pp_bind_rhs_buffer': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    'pp_solve_forward': (ctypes.c_int, [ctypes.c_void_p]),
    'pp_solve_forward_ex': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'pp_rs_buffer': (ctypes.c_void_p, [ctypes.c_void_p]),
    'pp_bind_rs_buffer': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    'pp_solve_coupling': (ctypes.c_int, [ctypes.c_void_p, _f64p]),
    'pp_solve_backward': (ctypes.c_int, [ctypes.c_void_p]),
    'pp_download_solution': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]),
    'pp_solution_buffer': (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int]),
    'pp_get_coupling_solution': (ctypes.c_int, [ctypes.c_void_p, _f64p]),
    'pp_increase_memory_allocation': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double]),
    'pp_set_memory_budget': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64]),
    'pp_memory_info': (ctypes.c_int, [ctypes.c_void_p, _i64p]),
    'pp_bcr_block_paths': (ctypes.c_int, [ctypes.c_void_p, _i32p]),
    'pp_synchronize': (ctypes.c_int, [ctypes.c_void_p]),
    'pp_profile': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'pp_phase_times': (ctypes.c_int, [ctypes.c_void_p, _f64p, _i32p, _i32p]),
    'pp_group_stats': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _i64p]),
    'pp_group_stats_ex': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _i64p]),
    'pp_group_perm': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _i32p]),
    'pp_set_diagonal_classes': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    'pp_numeric_local_shifted': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, ctypes.c_double]),
    'pp_stage_values': (ctypes.c_int, [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 8 + [ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                        ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    'pp_vec_step_stats': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 8 +
                          [ctypes.c_double, ctypes.c_double, _f64p]),
    'pp_vec_max_abs': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, _f64p]),
    'pp_vec_axpy': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]),
    'pp_vec_permute': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_int64, ctypes.c_int]),
    'pp_source_sha1': (ctypes.c_char_p, []),
    'pp_example_burgers_model': (ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_int] * 10 + [ctypes.c_double] * 4 + [ctypes.c_void_p] * 7),
    'pp_ip_rhs': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_double]),
    'pp_ip_step_lengths': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_double, ctypes.c_double,
                                          ctypes.c_void_p]),
    'pp_ip_take_step': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]),
    'pp_ip_residuals': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'pp_ip_publish': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_void_p]),
    'pp_ip_wait': (ctypes.c_int, [ctypes.c_void_p, _f64p]),
    'pp_ip_phase_times': (ctypes.c_int, [ctypes.c_void_p, _f64p, _i32p, _i32p]),
    'pp_comm_allgather': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]),
    'pp_comm_unique_id': (ctypes.c_int, [ctypes.POINTER(ctypes.c_uint8)]),
    'pp_comm_init': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint8)]),
    'pp_comm_size': (ctypes.c_int, [ctypes.c_void_p]),
    'pp_allreduce_schur': (ctypes.c_int, [ctypes.c_void_p]),
    'pp_allreduce_rs': (ctypes.c_int, [ctypes.c_void_p]),
    'pp_set_pivot_tolerance': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, ctypes.c_double]),
    'pp_get_growth_count': (ctypes.c_int, [ctypes.c_void_p, _i64p]),
    'pp_find_growth': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _i32p]),
    'pp_find_zero_pivot': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _i32p]),
    'pp_residual': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]),
    'pp_refine_solve_coupling': (ctypes.c_int, [ctypes.c_void_p]),
    'pp_residual_result': (ctypes.c_int, [ctypes.c_void_p, _f64p, ctypes.c_void_p]),
    'pp_refine_begin': (ctypes.c_int, [ctypes.c_void_p]),
    'pp_refine_end': (ctypes.c_int, [ctypes.c_void_p]),
    'pp_get_factor': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f64p, ctypes.c_int64]),
}



class IpGroup(ctypes.Structure):
    """pp_ip_group of include/parapint_hip.h (one pattern group of the interior-point step on the device)."""
    _fields_ = [(k, ctypes.c_int32) for k in ('n', 'mi', 'me', 'nfs', 'batch', 'bpad', 'src_dp', 'src_ds', 'nfw', 'ncz', 'obj_row', 'reserved')] + \
               [(k, ctypes.c_void_p) for k in ('W', 'bounds', 'data', 'src', 'G', 'rhs', 'delta', 'prog', 'terms', 'zoff')]


GROUP_STAT_KEYS = ['n', 'n_coupling', 'batch', 'n_pivots', 'n_2x2', 'n_levels', 'nnz_L', 'u_doubles',
                   'factor_fma', 'schur_fma', 'factor_tasks', 'update_runs', 'schur_tiles', 'schur_tile_records',
                   'canonical_entries', 'raw_entries']

_lib = None


def kernel_source_sha1():
    """SHA-1 over the kernel sources of the library (csrc/*.hip, *.hpp, *.cpp, in name order): ties measurements that are
    kept next to the code (profiles/pmc_traffic.json) to the build they were taken on."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(_HERE, 'csrc')
    for name in sorted(os.listdir(d)):
        if name.endswith(('.hip', '.hpp', '.cpp')):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), 'rb').read())
    return h.hexdigest()


def load_library():
    """Load libparapint_hip.so and declare every prototype.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (libamdhip64.so, SONAME libamdhip64.so.7); it must be the
    # one already mapped when our library is opened so the process holds a single HIP/HSA runtime
    # (our kernels run on torch's streams and RCCL buffers).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('parapint_amd: %s is not built (run __graft_entry__.build()); '
                           'there is no CPU fallback for the HIP solver' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if a declared symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    built_from = lib.pp_source_sha1().decode()
    if built_from != kernel_source_sha1():
        raise RuntimeError('parapint_amd: %s was built from other sources than the ones in %s (stamp %s): rebuild it with '
                           '__graft_entry__.build()' % (LIB_PATH, os.path.join(_HERE, 'csrc'), built_from))
    _lib = lib
    return lib


def i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


def f64(a):
    a = np.ascontiguousarray(a, dtype=np.double)
    return a, a.ctypes.data_as(_f64p)


class NativeSolver(object):
    """Thin object wrapper over one pp_handle."""

    def __init__(self, device=-1, stream=None):
        self.lib = load_library()
        h = ctypes.c_void_p()
        rc = self.lib.pp_create(ctypes.byref(h), int(device), ctypes.c_void_p(stream or 0))
        if rc != 0 or not h:
            msg = self.lib.pp_last_error(None)
            raise RuntimeError('parapint_amd: pp_create failed (status %d: %s): no usable HIP device; '
                               'the solver has no CPU fallback' % (rc, msg.decode() if msg else ''))
        self.h = h

    def close(self):
        if getattr(self, 'h', None):
            self.lib.pp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def error(self):
        msg = self.lib.pp_last_error(self.h)
        return msg.decode() if msg else ''

    def check(self, rc, what):
        if rc != 0:
            raise NativeError(rc, '%s failed with status %d: %s' % (what, rc, self.error()))

    def group_stats(self, group):
        out = np.zeros(16, dtype=np.int64)
        self.check(self.lib.pp_group_stats(self.h, group, out.ctypes.data_as(_i64p)), 'pp_group_stats')
        return dict(zip(GROUP_STAT_KEYS, [int(v) for v in out]))


    def group_stats_ex(self, group):
        out = np.zeros(16, dtype=np.int64)
        self.check(self.lib.pp_group_stats_ex(self.h, group, out.ctypes.data_as(_i64p)), 'pp_group_stats_ex')
        keys = ['raw_used', 'dinv_doubles', 'tm_doubles', 'coupling_entries', 'index_bytes', 'fwd_entries', 'crow_entries',
                'nsrc', 'launches_factor', 'launches_fwd', 'launches_bwd', 'bpad', 'nchunk', 'schur_tiles', 'tail_level0']
        return dict(zip(keys, [int(v) for v in out[:15]]))

    def get_factor(self, group, which, instance, count):
        out = np.zeros(int(count), dtype=np.double)
        self.check(self.lib.pp_get_factor(self.h, int(group), int(which), int(instance), out.ctypes.data_as(_f64p),
                                          ctypes.c_int64(int(count))), 'pp_get_factor')
        return out


class NativeError(RuntimeError):
    def __init__(self, status, msg):
        RuntimeError.__init__(self, msg)
        self.status = status
