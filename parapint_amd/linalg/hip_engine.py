"""The product engine of ``HipSchurComplementLinearSolver``: every numeric step runs in libparapint_hip.so on the GPU
(ctypes calls into the C ABI of include/parapint_hip.h; torch only for device memory and the current stream).  Split from
hip_schur_complement.py, which holds the host logic of the reference's class (mpi_explicit_schur_complement.py:128-452)."""
import ctypes
import zlib

import numpy as np

_S8 = (8,)


def _addr(a, _from_buffer=ctypes.c_char.from_buffer, _addressof=ctypes.addressof):
    """Address of the first element of a contiguous array (a third of the cost of ``a.ctypes.data``; the host boundary
    asks for two or three thousand of them per call)."""
    try:
        return _addressof(_from_buffer(a))
    except (TypeError, ValueError):          # read-only or empty buffer
        return a.ctypes.data


def _checksum(a):
    """CRC-32 over the bytes of an index array (position dependent: two entries exchanged in place change it)."""
    return zlib.crc32(a) if a.flags.c_contiguous else zlib.crc32(np.ascontiguousarray(a))


def _index_record(a):
    """What is remembered of an index array that was compared with a group's reference order: the object (kept alive,
    so that its id stays its own), size, address and checksum."""
    return (a, a.size, _addr(a), _checksum(a))


def _index_intact(a, rec, full):
    """The array recognised by identity still is what was verified: same size; and, whenever its address changed or a
    full check is due (every `pattern_check_interval`-th call), the same checksum.  A mismatch sends the block through
    the full comparison again (another entry order is canonicalised, entries outside the plan re-plan)."""
    if a.size != rec[1]:
        return False
    if full or _addr(a) != rec[2]:
        return _checksum(a) == rec[3]
    return True



class HipEngine(object):
    """The product engine: every numeric step runs in libparapint_hip.so on the GPU."""

    def __init__(self, device=None):
        import torch
        from parapint_amd import _native
        if not torch.cuda.is_available():
            raise RuntimeError('parapint_amd: no HIP device visible; the solver has no CPU fallback')
        self._torch = torch
        self._native = _native
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.stream = torch.cuda.current_stream(self.device).cuda_stream
        self.ns = _native.NativeSolver(self.device, self.stream)
        self.lib = self.ns.lib
        self.nc = 0
        self._S_t = None
        self._rs_t = None
        self._ip_ops = None
        self._resid_cpl = self._resid_out = self._resid_ptrs = None

    supports_block_tridiagonal = True

    def symbolic(self, nc, groups, btd=None, cinv=None):
        """nc: coupling dimension the library works with (padded to G * gs for a block-tridiagonal S); btd = (gs, G) or
        None; cinv: old -> new coupling order (block-tridiagonal S), applied to the maps of mapped groups."""
        import ctypes
        ns, lib, N = self.ns, self.lib, self._native
        self.nc = nc
        ns.check(lib.pp_begin_symbolic(ns.h, nc), 'pp_begin_symbolic')
        if btd is not None:
            ns.check(lib.pp_set_coupling_structure(ns.h, 1, int(btd[0]), int(btd[1])), 'pp_set_coupling_structure')
        for g in groups:
            keep = [N.i32(g.rowK), N.i32(g.colK), N.i32(g.rowB), N.i32(g.colB), N.i32(g.can_ptr), N.i32(g.can_idx)]
            rep = N.f64(g.rep_vals) if g.rep_vals is not None else (None, None)
            gid = ctypes.c_int(-1)
            cmaps = g.cmaps
            if cmaps and cmaps[0] is not None:
                cm = np.stack(cmaps).astype(np.int64) if g.m > 0 else np.zeros((len(cmaps), 0), dtype=np.int64)
                if cinv is not None:
                    cm = cinv[cm]
                cmk = N.i32(cm)
                ns.check(lib.pp_add_group_mapped(ns.h, g.n, len(g.blocks), g.rowK.size, keep[0][1], keep[1][1], g.rowB.size,
                                                 keep[2][1], keep[3][1], g.nraw, keep[4][1], keep[5][1], rep[1], int(g.m),
                                                 cmk[1], ctypes.byref(gid)), 'pp_add_group_mapped')
            else:
                ns.check(lib.pp_add_group(ns.h, g.n, len(g.blocks), g.rowK.size, keep[0][1], keep[1][1], g.rowB.size,
                                          keep[2][1], keep[3][1], g.nraw, keep[4][1], keep[5][1], rep[1],
                                          ctypes.byref(gid)), 'pp_add_group')
        ns.check(lib.pp_end_symbolic(ns.h), 'pp_end_symbolic')
        torch = self._torch
        dev = torch.device('cuda', self.device)
        self.schur_doubles = int(lib.pp_schur_buffer_doubles(ns.h)) - 8
        self._S_t = torch.zeros(self.schur_doubles + 8, dtype=torch.float64, device=dev)   # S | status / inertia / growth tail
        self._rs_t = torch.zeros(max(nc, 1), dtype=torch.float64, device=dev)
        ns.check(lib.pp_bind_schur_buffer(ns.h, self._S_t.data_ptr()), 'pp_bind_schur_buffer')
        ns.check(lib.pp_bind_rs_buffer(ns.h, self._rs_t.data_ptr()), 'pp_bind_rs_buffer')
        return [ns.group_stats(i) for i in range(len(groups))]

    def ip_ops(self):
        """The kernels of the interior-point step on device-resident iterates (parapint_amd.linalg.device_ip_ops)."""
        if self._ip_ops is None:
            from parapint_amd.linalg.device_ip_ops import HipIpOps
            self._ip_ops = HipIpOps(self)
        return self._ip_ops

    def get_factor(self, gid, which, instance, count):
        """Diagnostic: factor storage of one block (0 = U panels, 1 = L rows, 2 = pivot inverses)."""
        return self.ns.get_factor(gid, which, instance, count)

    def find_zero_pivot(self, gid):
        """Slot of the first instance of group gid whose block hit a numerically zero pivot in the last numeric
        factorisation, or -1."""
        import ctypes
        out = ctypes.c_int32(-1)
        self.ns.check(self.lib.pp_find_zero_pivot(self.ns.h, gid, ctypes.byref(out)), 'pp_find_zero_pivot')
        return int(out.value)

    def alloc_pinned(self, shape):
        """Page-locked host array (zero-filled): H2D / D2H copies from it are asynchronous and run at PCIe speed.  The
        memory is released when the last view of the array is gone."""
        import ctypes
        import weakref
        n = int(np.prod(shape))
        ptr = self.lib.pp_host_alloc(ctypes.c_int64(8 * n)) if n > 0 else None
        if not ptr:
            return np.zeros(shape, dtype=np.double)          # pageable: slower, still correct
        buf = (ctypes.c_double * n).from_address(ptr)
        weakref.finalize(buf, self.lib.pp_host_free, ptr)
        arr = np.frombuffer(buf, dtype=np.double).reshape(shape)
        arr[...] = 0.0
        return arr

    def find_growth(self, gid):
        """Slot of the first instance of group gid whose factor exceeded the growth bound 1 / u_runtime, or -1."""
        import ctypes
        out = ctypes.c_int32(-1)
        self.ns.check(self.lib.pp_find_growth(self.ns.h, gid, ctypes.byref(out)), 'pp_find_growth')
        return int(out.value)

    def growth_count(self):
        """Instances (all ranks) whose last factorisation produced a factor entry beyond the growth bound."""
        import ctypes
        out = ctypes.c_int64(0)
        self.ns.check(self.lib.pp_get_growth_count(self.ns.h, ctypes.byref(out)), 'pp_get_growth_count')
        return int(out.value)

    def set_pivot_tolerance(self, u_symbolic, u_runtime):
        self.ns.check(self.lib.pp_set_pivot_tolerance(self.ns.h, float(u_symbolic), float(u_runtime)),
                      'pp_set_pivot_tolerance')

    def stage_upload(self, g, items, full_check=True):
        """items: [(slot, (kr, kc, kd, br, bc, bd))] of one pattern group, ascending slots (int32 / float64 arrays,
        contiguous).  The needed runs of every block that is in the group's reference entry order go to its compact
        staging row and on to the device, slice by slice, with the copies overlapping the staging of the next slice
        (include/parapint_hip.h: pp_stage_upload_compact); returns one flag per item (False: the caller stages and
        uploads that block itself)."""
        import os
        n = len(items)
        cols = ([], [], [], [], [], [])
        nnzk, nnzb, slots = [], [], []
        ref = g._ref32
        if ref is None:
            ref = g._ref32 = [np.ascontiguousarray(r, dtype=np.int32) for r in g.raw_refs]
            g._refptr = [r.ctypes.data for r in ref]
        known, refptr = g.known_ptrs, g._refptr           # (role, id(index array)) -> the array (kept alive), verified equal to the reference
        unknown = []
        memo = {}
        for i, (slot, arrays) in enumerate(items):
            # index arrays already verified against the reference order (the same objects as at an earlier call): hand
            # the library the reference pointers themselves, so that it skips the comparison
            fresh = False
            for q, r in ((0, 0), (1, 1), (3, 2), (4, 3)):
                a = arrays[q]
                # recognised only in the ROLE it was verified in (K rows / K columns / border rows / border columns), with
                # its size, address and -- every k-th call -- checksum unchanged: an array object that was mutated in
                # place or is reused in another role goes through the full comparison again
                k = known.get((q, id(a)))
                if k is not None:
                    hit = memo.get((q, id(a)))
                    if hit is None:
                        # (blocks arrive here when they are new or were just found changed: always the checksum)
                        hit = memo[(q, id(a))] = _index_intact(a, k, True)
                    if not hit:
                        k = None
                if k is not None:
                    cols[q].append(refptr[r])
                else:
                    cols[q].append(_addr(a))
                    fresh = True
            if fresh:
                unknown.append(i)
            kd, bd = arrays[2], arrays[5]
            cols[2].append(_addr(kd))
            cols[5].append(_addr(bd) if bd.size else 0)
            nnzk.append(kd.size)
            nnzb.append(bd.size)
            slots.append(slot)
        ptr = np.array(cols, dtype=np.uint64)
        nnz = np.array((nnzk, nnzb), dtype=np.int64)
        slots = np.array(slots, dtype=np.int32)
        same = np.zeros(n, dtype=np.uint8)
        rk, rb = g.runsK, g.runsB
        rc = self.lib.pp_stage_upload_compact(self.ns.h, g.gid, n, min(16, os.cpu_count() or 1), ptr[0].ctypes.data,
                                              ptr[1].ctypes.data, ptr[2].ctypes.data, nnz[0].ctypes.data, ptr[3].ctypes.data,
                                              ptr[4].ctypes.data, ptr[5].ctypes.data, nnz[1].ctypes.data, ref[0].ctypes.data,
                                              ref[1].ctypes.data, ctypes.c_int64(g.nrawK), ref[2].ctypes.data,
                                              ref[3].ctypes.data, ctypes.c_int64(g.nraw - g.nrawK), rk.shape[0],
                                              rk.ctypes.data, rb.shape[0], rb.ctypes.data, g.staging.ctypes.data,
                                              slots.ctypes.data, same.ctypes.data)
        self.ns.check(rc, 'pp_stage_upload_compact')
        ok = same.astype(bool)
        if len(known) < 4 * n + 64:           # (one generation of index arrays at most is kept alive)
            for i in unknown:
                if ok[i]:
                    for q in (0, 1, 3, 4):
                        a = items[i][1][q]
                        if (q, id(a)) not in known or not memo.get((q, id(a)), False):
                            known[(q, id(a))] = _index_record(a)
                            memo[(q, id(a))] = True
        return ok

    def stage_upload_verified(self, g, slots, kd_ptr, bd_ptr, runs=None):
        """The same for blocks whose index arrays are the very objects verified at an earlier call (ascending slots,
        addresses of their K and border data): nothing is compared, every block is staged and uploaded.  Returns at
        once -- the library's host threads work while the caller prepares its next batch; stage_upload_end() waits.
        runs: (runsK, runsB) other than the group's -- a subset of them, for rows whose other entries the staging row
        already holds (solver.declare_constant_entries)."""
        import os
        kd = np.array(kd_ptr, dtype=np.uint64)
        bd = np.array(bd_ptr, dtype=np.uint64)
        sl = np.array(slots, dtype=np.int32)
        rk, rb = (g.runsK, g.runsB) if runs is None else (np.ascontiguousarray(runs[0], dtype=np.int64),
                                                         np.ascontiguousarray(runs[1], dtype=np.int64))
        rc = self.lib.pp_stage_upload_verified_begin(self.ns.h, g.gid, len(slots), min(16, os.cpu_count() or 1), kd.ctypes.data,
                                                     bd.ctypes.data, ctypes.c_int64(g.nrawK), ctypes.c_int64(g.nraw - g.nrawK),
                                                     rk.shape[0], rk.ctypes.data, rb.shape[0], rb.ctypes.data,
                                                     g.staging.ctypes.data, sl.ctypes.data)
        self.ns.check(rc, 'pp_stage_upload_verified_begin')

    def stage_upload_end(self):
        self.ns.check(self.lib.pp_stage_upload_end(self.ns.h), 'pp_stage_upload_end')

    def upload_rhs_rows(self, g, vectors):
        """vectors: one contiguous float64 vector of g.n entries per block of the group, slot order; through the pinned
        staging array to the device, slice by slice (include/parapint_hip.h: pp_upload_rhs_rows)."""
        import os
        src = np.array([_addr(v) for v in vectors], dtype=np.uint64)
        self.ns.check(self.lib.pp_upload_rhs_rows(self.ns.h, g.gid, len(vectors), min(16, os.cpu_count() or 1),
                                                  src.ctypes.data, g.rhs_staging.ctypes.data), 'pp_upload_rhs_rows')

    def download_solution_rows(self, g, pinned, out=None):
        """Solutions of a group into the pinned array (asynchronous: synchronize() before reading), or through it
        into ``out`` (pageable, complete on return)."""
        import os
        self.ns.check(self.lib.pp_download_solution_rows(self.ns.h, g.gid, min(16, os.cpu_count() or 1), pinned.ctypes.data,
                                                         None if out is None else out.ctypes.data),
                      'pp_download_solution_rows')

    def copy_rows(self, dst, rows):
        """rows: [(row index of dst, contiguous float64 vector of dst.shape[1] entries)] copied on host threads."""
        import os
        n = len(rows)
        src = np.empty(n, dtype=np.uint64)
        idx = np.empty(n, dtype=np.int64)
        for i, (r, v) in enumerate(rows):
            src[i] = v.ctypes.data
            idx[i] = r
        rc = self.lib.pp_copy_rows(n, min(16, os.cpu_count() or 1), src.ctypes.data, idx.ctypes.data, dst.ctypes.data,
                                   dst.shape[1])
        if rc != 0:
            raise RuntimeError('pp_copy_rows failed with status %d' % rc)

    def upload_values_compact(self, gid, staging, row0=0, nrows=None):
        nrows = staging.shape[0] - row0 if nrows is None else nrows
        self.ns.check(self.lib.pp_upload_values_compact(self.ns.h, gid, staging.ctypes.data, int(row0), int(nrows), 0),
                      'pp_upload_values_compact')

    def set_value_map(self, gid, nsrc, src, coef):
        s32 = np.ascontiguousarray(src, dtype=np.int32)
        c64 = np.ascontiguousarray(coef, dtype=np.double)
        import ctypes
        self.ns.check(self.lib.pp_set_value_map(self.ns.h, gid, int(nsrc), s32.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                                c64.ctypes.data_as(ctypes.POINTER(ctypes.c_double))), 'pp_set_value_map')

    def new_tensor(self, shape):
        torch = self._torch
        return torch.zeros(shape, dtype=torch.float64, device=torch.device('cuda', self.device))

    def new_tensor_uninitialized(self, shape):
        torch = self._torch
        return torch.empty(shape, dtype=torch.float64, device=torch.device('cuda', self.device))

    def bind_source_tensor(self, gid, tensor):
        self.ns.check(self.lib.pp_bind_source_buffer(self.ns.h, gid, tensor.data_ptr()), 'pp_bind_source_buffer')

    def bind_rhs_tensor(self, gid, tensor):
        self.ns.check(self.lib.pp_bind_rhs_buffer(self.ns.h, gid, tensor.data_ptr() if tensor is not None else None),
                      'pp_bind_rhs_buffer')

    def bind_solution_tensor(self, gid, tensor):
        self.ns.check(self.lib.pp_bind_solution_buffer(self.ns.h, gid, tensor.data_ptr() if tensor is not None else None),
                      'pp_bind_solution_buffer')

    def bind_native_vectors(self, gid, rhs, x):
        """[n][padded batch] device tensors the sweeps read b from / write x to (None, None: back to [batch][n] copies)."""
        for t in (rhs, x):
            if t is not None and (not t.is_contiguous() or not t.is_cuda or str(t.dtype) != 'torch.float64'):
                raise ValueError('native vectors must be contiguous float64 device tensors')
        self.ns.check(self.lib.pp_bind_native_vectors(self.ns.h, gid, rhs.data_ptr() if rhs is not None else None,
                                                      x.data_ptr() if x is not None else None), 'pp_bind_native_vectors')

    def solve_coupling_dev(self, tensor):
        self.ns.check(self.lib.pp_solve_coupling_dev(self.ns.h, tensor.data_ptr() if tensor is not None else None),
                      'pp_solve_coupling_dev')

    def copy_coupling_solution(self, tensor):
        self.ns.check(self.lib.pp_copy_coupling_solution(self.ns.h, tensor.data_ptr()), 'pp_copy_coupling_solution')

    def index_tensor(self, idx):
        """int64 index array on the device (for permute)."""
        torch = self._torch
        return torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int64)).to(torch.device('cuda', self.device))

    def permute(self, idx_t, src, dst, scatter):
        """scatter: dst = 0, dst[idx] = src; else dst = src[idx] (stream-ordered on the solver's stream)."""
        import ctypes
        n = int(idx_t.numel())
        if (src.numel() if scatter else dst.numel()) < n or not src.is_contiguous() or not dst.is_contiguous():
            raise ValueError('permute: operands do not match the index array')
        self.ns.check(self.lib.pp_vec_permute(self.ns.h, ctypes.c_int64(n), idx_t.data_ptr(), src.data_ptr(), dst.data_ptr(),
                                              ctypes.c_int64(dst.numel()), 1 if scatter else 0), 'pp_vec_permute')

    def upload_values(self, gid, raw):
        self.ns.check(self.lib.pp_upload_values(self.ns.h, gid, raw.ctypes.data, 0), 'pp_upload_values')

    def upload_values_device(self, gid, tensor):
        self.ns.check(self.lib.pp_upload_values(self.ns.h, gid, tensor.data_ptr(), 1), 'pp_upload_values')

    def set_diagonal_classes(self, gid, cls):
        c = np.ascontiguousarray(cls, dtype=np.int8)
        self.ns.check(self.lib.pp_set_diagonal_classes(self.ns.h, gid, c.ctypes.data), 'pp_set_diagonal_classes')

    def numeric_local_shifted(self, delta_w, delta_c):
        self.ns.check(self.lib.pp_numeric_local_shifted(self.ns.h, float(delta_w), float(delta_c)),
                      'pp_numeric_local_shifted')

    def numeric_local(self):
        self.ns.check(self.lib.pp_numeric_local(self.ns.h), 'pp_numeric_local')

    def numeric_factor_blocks(self):
        self.ns.check(self.lib.pp_numeric_factor_blocks(self.ns.h), 'pp_numeric_factor_blocks')

    def numeric_schur(self, side=False):
        """side: on the library's own stream behind the factor levels (a forward sweep enqueued afterwards overlaps it)."""
        self.ns.check(self.lib.pp_numeric_schur_ex(self.ns.h, 1 if side else 0), 'pp_numeric_schur')

    def fail_local(self, status):
        self.ns.check(self.lib.pp_fail_local(self.ns.h, int(status)), 'pp_fail_local')

    def set_memory_budget(self, nbytes):
        self.ns.check(self.lib.pp_set_memory_budget(self.ns.h, int(nbytes)), 'pp_set_memory_budget')

    def memory_info(self):
        """(bytes of device value storage the plan needs at most, effective budget or 0, bytes allocated now)"""
        import ctypes
        out = np.zeros(3, dtype=np.int64)
        self.ns.check(self.lib.pp_memory_info(self.ns.h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))),
                      'pp_memory_info')
        return int(out[0]), int(out[1]), int(out[2])

    def bcr_block_paths(self):
        """Block-tridiagonal S: (diagonal blocks inverted from the unpivoted LDL^T, blocks left to Bunch-Kaufman) of the
        last factorisation of S; (0, 0) for a dense S.  Diagnostic, synchronises."""
        import ctypes
        out = np.zeros(2, dtype=np.int32)
        self.ns.check(self.lib.pp_bcr_block_paths(self.ns.h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))),
                      'pp_bcr_block_paths')
        return int(out[0]), int(out[1])

    def _direct_rccl(self, comm):
        """The two data-path all-reduces (and the all-gathers of the interior-point step) are enqueued by the library itself
        as RCCL calls on the handle's stream (include/parapint_hip.h: pp_allreduce_schur / pp_allreduce_rs /
        pp_comm_allgather) instead of by torch.distributed between the kernel enqueues.  Round 5: the DEFAULT for a
        communicator with device collectives and at least two ranks (PP_DIRECT_RCCL=0 or comm.direct_rccl = False keeps the
        torch.distributed calls; PP_DIRECT_RCCL=1 / comm.direct_rccl = True also takes a one-rank group through RCCL).  The
        communicator is made once per handle from a unique id that rank 0 broadcasts through the torch process group; if
        librccl cannot be opened the torch path is used."""
        import os
        if not getattr(comm, 'device_collectives', False):
            return False
        want = getattr(comm, 'direct_rccl', None)
        env = os.environ.get('PP_DIRECT_RCCL')
        if want is False or env == '0':
            return False
        if not (want is True or env == '1' or comm.size >= 2):
            return False
        if getattr(self, '_rccl_unavailable', False):
            return False
        if self.lib.pp_comm_size(self.ns.h) != comm.size:
            import ctypes
            torch = self._torch
            uid = np.zeros(128, dtype=np.uint8)
            # EVERY rank first finds out whether it can open librccl at all (an id of its own, thrown away) and the ranks
            # agree on that through the torch group: ncclCommInitRank is collective -- a rank that cannot call it would
            # leave the others waiting inside it, where no later flag can reach them
            ok = 1 if self.lib.pp_comm_unique_id(uid.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))) == 0 else 0
            if comm.size > 1:
                can = torch.tensor([ok], dtype=torch.int32, device='cuda')
                comm._dist.all_reduce(can, op=comm._dist.ReduceOp.MIN, group=comm._group)
                ok = int(can.item())
            if not ok:
                if want is True or env == '1':
                    raise RuntimeError('pp_comm_unique_id failed on some rank (librccl not available)')
                self._rccl_unavailable = True
                return False
            # (rank 0's id is the communicator's)
            t = torch.from_numpy(np.concatenate([uid, np.array([ok], dtype=np.uint8)])).cuda()
            if comm.size > 1:
                comm._dist.broadcast(t, src=0, group=comm._group)
            got = t.cpu().numpy()
            uid = np.ascontiguousarray(got[:128])
            ok = 1
            try:
                self.ns.check(self.lib.pp_comm_init(self.ns.h, int(comm.size), int(comm.rank),
                                                    uid.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))), 'pp_comm_init')
                # one all-gather of the rank numbers through the new communicator: it must return 0 .. size - 1
                mine = torch.full((1,), float(comm.rank), dtype=torch.float64, device='cuda')
                table = torch.full((comm.size,), -1.0, dtype=torch.float64, device='cuda')
                torch.cuda.synchronize()          # (the two fills ran on torch's stream, the gather runs on the handle's)
                self.ns.check(self.lib.pp_comm_allgather(self.ns.h, mine.data_ptr(), table.data_ptr(), ctypes.c_int64(1)),
                              'pp_comm_allgather')
                self.ns.check(self.lib.pp_synchronize(self.ns.h), 'pp_synchronize')
                if not bool((table.cpu() == torch.arange(comm.size, dtype=torch.float64)).all()):
                    ok = 0
            except Exception:
                if want is True or env == '1':
                    raise
                ok = 0
            if comm.size > 1:
                # (all ranks take the same path: one rank's failure sends everybody to torch.distributed)
                flag = torch.tensor([ok], dtype=torch.int32, device='cuda')
                comm._dist.all_reduce(flag, op=comm._dist.ReduceOp.MIN, group=comm._group)
                ok = int(flag.item())
            if not ok:
                self._rccl_unavailable = True
                return False
        return True

    def allreduce_schur(self, comm):
        if comm.size > 1 or getattr(comm, 'always_reduce', False):
            if self._direct_rccl(comm):
                self.ns.check(self.lib.pp_allreduce_schur(self.ns.h), 'pp_allreduce_schur')
            elif comm.device_collectives:
                comm.allreduce_sum_tensor_(self._S_t)
            else:
                host = comm.allreduce_sum(self._S_t.cpu().numpy())
                self._S_t.copy_(self._torch.from_numpy(host))

    def factor_schur(self, Q):
        if Q is None:
            self.ns.check(self.lib.pp_factor_schur(self.ns.h, None), 'pp_factor_schur')
        else:
            Qf, Qp = self._native.f64(np.asfortranarray(Q).ravel(order='F'))
            self.ns.check(self.lib.pp_factor_schur(self.ns.h, Qp), 'pp_factor_schur')
            self.ns.check(self.lib.pp_synchronize(self.ns.h), 'pp_synchronize')   # Qf must outlive the H2D

    def factor_schur_flat(self, Qflat):
        """Q in the layout of the Schur buffer (block-tridiagonal S), or None."""
        if Qflat is None:
            self.ns.check(self.lib.pp_factor_schur(self.ns.h, None), 'pp_factor_schur')
        else:
            Qf, Qp = self._native.f64(Qflat)
            self.ns.check(self.lib.pp_factor_schur(self.ns.h, Qp), 'pp_factor_schur')
            self.ns.check(self.lib.pp_synchronize(self.ns.h), 'pp_synchronize')

    def factor_schur_corner(self, pos, val):
        """Block-tridiagonal S: Q as (position in the Schur layout, value) pairs."""
        pos = np.ascontiguousarray(pos, dtype=np.int64)
        val = np.ascontiguousarray(val, dtype=np.float64)
        import ctypes
        self.ns.check(self.lib.pp_factor_schur_corner(self.ns.h, int(pos.size), pos.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                                      val.ctypes.data_as(ctypes.POINTER(ctypes.c_double))), 'pp_factor_schur_corner')

    def set_coupling_schedule(self, sequential):
        self.ns.check(self.lib.pp_set_coupling_schedule(self.ns.h, 1 if sequential else 0), 'pp_set_coupling_schedule')

    def get_schur_flat(self):
        S = np.zeros(self.schur_doubles)
        _, p = self._native.f64(S)
        self.ns.check(self.lib.pp_get_schur(self.ns.h, S.ctypes.data_as(type(p))), 'pp_get_schur')
        return S

    def set_supernodes(self, wmax, tol_rows):
        """Block-pivot merging for the next symbolic factorisation (0 / -1: library defaults)."""
        self.ns.check(self.lib.pp_set_supernodes(self.ns.h, int(wmax), int(tol_rows)), 'pp_set_supernodes')

    def set_dense_policy(self, policy):
        """0: optimistic blocked LDL^T (fp64 MFMA) with Bunch-Kaufman fallback; 1: Bunch-Kaufman only."""
        self.ns.check(self.lib.pp_set_dense_policy(self.ns.h, int(policy)), 'pp_set_dense_policy')

    def dense_mode(self):
        """1 if the last S factorisation was the accepted blocked LDL^T, 0 if Bunch-Kaufman."""
        import ctypes
        m = ctypes.c_int(-1)
        self.ns.check(self.lib.pp_get_dense_mode(self.ns.h, ctypes.byref(m)), 'pp_get_dense_mode')
        return int(m.value)

    def status(self):
        out = np.zeros(4, dtype=np.int64)
        import ctypes
        self.ns.check(self.lib.pp_get_status(self.ns.h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))),
                      'pp_get_status')
        return int(out[0]), int(out[1]), int(out[2]), int(out[3])

    def get_schur(self):
        S = np.zeros(self.nc * self.nc)
        _, p = self._native.f64(S)
        self.ns.check(self.lib.pp_get_schur(self.ns.h, S.ctypes.data_as(type(p))), 'pp_get_schur')
        return S.reshape((self.nc, self.nc), order='F')

    def upload_rhs(self, gid, rhs):
        self.ns.check(self.lib.pp_upload_rhs(self.ns.h, gid, rhs.ctypes.data, 0), 'pp_upload_rhs')

    def solve_forward(self, early=False):
        """early: the bound right-hand side was complete before this step's factorisation was enqueued (pp_solve_forward_ex)."""
        self.ns.check(self.lib.pp_solve_forward_ex(self.ns.h, 1 if early else 0), 'pp_solve_forward')

    def allreduce_rs(self, comm):
        if comm.size > 1 or getattr(comm, 'always_reduce', False):
            if self._direct_rccl(comm):
                self.ns.check(self.lib.pp_allreduce_rs(self.ns.h), 'pp_allreduce_rs')
            elif comm.device_collectives:
                comm.allreduce_sum_tensor_(self._rs_t)
            else:
                host = comm.allreduce_sum(self._rs_t.cpu().numpy())
                self._rs_t.copy_(self._torch.from_numpy(host))

    def solve_coupling(self, rc):
        if rc is None:
            self.ns.check(self.lib.pp_solve_coupling(self.ns.h, None), 'pp_solve_coupling')
        else:
            rcf, rcp = self._native.f64(rc)
            self.ns.check(self.lib.pp_solve_coupling(self.ns.h, rcp), 'pp_solve_coupling')
            self.ns.check(self.lib.pp_synchronize(self.ns.h), 'pp_synchronize')

    def solve_backward(self):
        self.ns.check(self.lib.pp_solve_backward(self.ns.h), 'pp_solve_backward')

    def download_solution(self, gid, out):
        self.ns.check(self.lib.pp_download_solution(self.ns.h, gid, out.ctypes.data, 0), 'pp_download_solution')

    def coupling_solution(self):
        xc = np.zeros(max(self.nc, 1))
        _, p = self._native.f64(xc)
        self.ns.check(self.lib.pp_get_coupling_solution(self.ns.h, xc.ctypes.data_as(type(p))),
                      'pp_get_coupling_solution')
        return xc[:self.nc]

    def residual_begin(self, store=False, bc=None, on_device=False):
        """Enqueues the a-posteriori check of the last back-solve (include/parapint_hip.h: pp_residual) behind it on the
        solver's stream; residual_end() waits for its result.  bc: the coupling right-hand side as a device tensor, or None
        (the one of the last coupling solve).  on_device: 1 -- the sums of the coupling rows are complete on this rank; 2 --
        several ranks, the sums and the block results meet in one all-reduce of the library's communicator (collective)."""
        self.ns.check(self.lib.pp_residual(self.ns.h, 1 if store else 0, bc.data_ptr() if bc is not None else None,
                                           int(on_device)), 'pp_residual')

    def residual_end(self):
        """(rho of the worst local instance, its group, its slot, largest row scale of the blocks, rho of the coupling rows --
        and, on_device = 2, of the worst block of any rank -- or None, x_c, sum_i A_i x_i, sum_i |A_i||x_i|, b_c): the four
        coupling vectors (library order) only when the caller has to finish the coupling rows itself (rho None)."""
        nc = self.nc
        out = self._resid_out
        cpl = self._resid_cpl
        if cpl is None or cpl.size != 4 * max(nc, 1):
            cpl = self._resid_cpl = np.zeros(4 * max(nc, 1))
            out = self._resid_out = np.zeros(6)
            self._resid_ptrs = (out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), cpl.ctypes.data)
        self.ns.check(self.lib.pp_residual_result(self.ns.h, self._resid_ptrs[0], self._resid_ptrs[1]), 'pp_residual_result')
        if out[4] >= 0.0 or out[4] != out[4]:
            # (rho of the coupling rows AND of the worst block of any rank: max of the two is the verdict, the same on every rank)
            rho_rows = float(out[4]) if out[4] == out[4] else np.inf
            rho_all = float(out[5]) if out[5] == out[5] else np.inf
            return float(out[0]), int(out[1]), int(out[2]), float(out[3]), max(rho_rows, rho_all), None, None, None, None
        return (float(out[0]), int(out[1]), int(out[2]), float(out[3]), None, cpl[:nc].copy(), cpl[nc:2 * nc].copy(),
                cpl[2 * nc:3 * nc].copy(), cpl[3 * nc:4 * nc].copy())

    def residual(self, store=False, bc=None, on_device=False):
        self.residual_begin(store, bc, on_device)
        return self.residual_end()

    def refine_solve_coupling(self):
        self.ns.check(self.lib.pp_refine_solve_coupling(self.ns.h), 'pp_refine_solve_coupling')

    def refine_begin(self):
        self.ns.check(self.lib.pp_refine_begin(self.ns.h), 'pp_refine_begin')

    def refine_end(self):
        self.ns.check(self.lib.pp_refine_end(self.ns.h), 'pp_refine_end')

    def synchronize(self):
        self.ns.check(self.lib.pp_synchronize(self.ns.h), 'pp_synchronize')

    def increase_memory_allocation(self, factor):
        self.ns.check(self.lib.pp_increase_memory_allocation(self.ns.h, float(factor)), 'pp_increase_memory_allocation')
