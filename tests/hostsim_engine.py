"""TEST-ONLY numeric engine for HipSchurComplementLinearSolver built on the host interpreter
(tests/hostsim).  It lets the CPU suite rehearse the solver class's host logic -- grouping,
canonicalisation (quirk Q7), ownership, the two all-reduces, status agreement -- on machines
without a GPU, including world_size-2 gloo runs.  The product never constructs it."""
import ctypes

import numpy as np

import hostsim_util as hu


class _SimGroup(object):
    pass


class _StatusError(RuntimeError):
    """Mirrors parapint_amd._native.NativeError: an exception that carries a LinearSolverStatus value."""

    def __init__(self, status, msg):
        RuntimeError.__init__(self, msg)
        self.status = status


def hu_status_error(status, msg):
    return _StatusError(status, msg)


class HostSimEngine(object):
    def __init__(self):
        self.groups = []
        self.nc = 0
        self.budget = None
        self.mem_factor = 1.0
        hu.lib().ppsim_set_pivot_tolerance(ctypes.c_double(0.0), ctypes.c_double(0.0))   # (process-wide in the interpreter: a new engine starts from the defaults)

    supports_block_tridiagonal = True     # (kept as a dense matrix in the permuted, padded ordering: the host logic --
                                          # ordering, padding, flat layouts, the clique table all-reduce -- is the same)

    def symbolic(self, nc, groups, btd=None, cinv=None):
        L = hu.lib()
        self.nc = nc
        self.btd = btd
        self.groups = []
        stats = []
        for g in groups:
            sg = _SimGroup()
            sg.g = g
            rep = None if g.rep_vals is None else np.ascontiguousarray(g.rep_vals, dtype=np.double)
            sg.keep = [np.ascontiguousarray(a, dtype=np.int32) for a in (g.rowK, g.colK, g.rowB, g.colB)]
            sg.m = getattr(g, 'm', nc)
            sg.cmaps = getattr(g, 'cmaps', None)
            if not sg.cmaps or sg.cmaps[0] is None:
                sg.cmaps = None
            elif cinv is not None:
                sg.cmaps = [cinv[np.asarray(cm, dtype=np.int64)] for cm in sg.cmaps]
            L.ppsim_set_batch_hint(len(g.blocks))            # the task sizes the device library picks for this batch
            L.ppsim_set_mapped_hint(1 if sg.cmaps is not None else 0)     # ... and its choice among elimination orders
            sg.h = ctypes.c_void_p(L.ppsim_create(g.n, sg.m, g.rowK.size, hu._ip(sg.keep[0]), hu._ip(sg.keep[1]),
                                                  g.rowB.size, hu._ip(sg.keep[2]), hu._ip(sg.keep[3]),
                                                  None if rep is None else hu._dp(rep), 0, -1, ctypes.c_double(-1.0)))
            L.ppsim_set_batch_hint(0)
            L.ppsim_set_mapped_hint(0)
            err = L.ppsim_error(sg.h)
            if err:
                raise RuntimeError(err.decode())
            st = np.zeros(13, dtype=np.int64)
            L.ppsim_stats(sg.h, st.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
            sg.usize, sg.npiv = int(st[5]), int(st[2])
            sg.batch = len(g.blocks)
            sg.raw = None
            self.groups.append(sg)
            stats.append({'n': int(st[0]), 'n_pivots': int(st[2]), 'n_levels': int(st[3]), 'n_2x2': int(st[4]) % 1000000,
                          'nnz_L': int(st[6])})
        self.S = np.zeros((nc, nc))
        self.tail = np.zeros(8)
        return stats

    def upload_values(self, gid, raw):
        self.groups[gid].raw = np.array(raw, dtype=np.double, copy=True)

    def upload_values_compact(self, gid, compact, row0=0, nrows=None):
        """Compact rows (only the raw entries some canonical entry reads, solver._Group.used) -> full raw rows."""
        sg = self.groups[gid]
        g = sg.g
        if sg.raw is None or sg.raw.shape != (sg.batch, g.nraw):
            sg.raw = np.zeros((sg.batch, g.nraw))
        nrows = compact.shape[0] - row0 if nrows is None else nrows
        sg.raw[row0:row0 + nrows][:, g.used] = compact[row0:row0 + nrows]

    def numeric_local(self):
        L = hu.lib()
        nc = self.nc
        self.S = np.zeros((nc, nc))
        inertia = np.zeros(3, dtype=np.int64)
        growth = 0
        for sg in self.groups:
            g = sg.g
            sg.U, sg.Dinv, sg.L = [], [], []
            sg.can = []                   # canonical values of the factorisation (the residual check reads them)
            sg.zero_slot = -1
            sg.growth_slot = -1
            for b in range(sg.batch):
                zeros_before = int(inertia[2])
                can = np.add.reduceat(sg.raw[b][g.can_idx], g.can_ptr[:-1]) if g.can_idx.size else np.zeros(0)
                U = np.zeros(sg.usize)
                Lf = np.zeros(sg.usize)
                D = np.zeros(L.ppsim_dsize(sg.h))
                Sb = np.zeros((sg.m, sg.m))
                L.ppsim_factor(sg.h, hu._dp(np.ascontiguousarray(can)), hu._dp(U), hu._dp(Lf), hu._dp(D), hu._dp(Sb),
                               inertia.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), ctypes.c_double(1e-13))
                if int(inertia[2]) > zeros_before and sg.zero_slot < 0:
                    sg.zero_slot = b
                if L.ppsim_last_growth():
                    growth += 1
                    if sg.growth_slot < 0:
                        sg.growth_slot = b
                Sb = np.tril(Sb) + np.tril(Sb, -1).T
                if sg.cmaps is None:
                    self.S += Sb
                else:
                    cm = sg.cmaps[b]
                    self.S[np.ix_(cm, cm)] += Sb
                sg.U.append(U)
                sg.L.append(Lf)
                sg.Dinv.append(D)
                sg.can.append(np.array(can, dtype=np.double))
        self.tail = np.array([inertia[2], inertia[0], inertia[1], 0.0, growth, 0.0, 0.0, 0.0], dtype=np.double)

    def numeric_factor_blocks(self):
        """(the interpreter forms the factor and the Schur contribution in one pass)"""
        if self.budget is not None and self.required_bytes() > self.budget * self.mem_factor:
            raise hu_status_error(1, 'value storage exceeds the budget')
        self.numeric_local()

    def numeric_schur(self, side=False):
        pass

    def fail_local(self, status):
        self.S = np.zeros((self.nc, self.nc))
        self.tail = np.array([0.0, 0.0, 0.0, {1: 1.0, 2: 1e3, 3: 1e6}[int(status)], 0.0, 0.0, 0.0, 0.0])

    def required_bytes(self):
        return sum(8 * 2 * sg.usize * sg.batch for sg in self.groups)

    def set_memory_budget(self, nbytes):
        self.budget = int(nbytes)
        self.mem_factor = 1.0

    def memory_info(self):
        return self.required_bytes(), 0 if self.budget is None else int(self.budget * self.mem_factor), 0

    def growth_count(self):
        return int(round(self.tail[4]))

    def find_growth(self, gid):
        return self.groups[gid].growth_slot

    def set_pivot_tolerance(self, u_symbolic, u_runtime):
        hu.lib().ppsim_set_pivot_tolerance(ctypes.c_double(float(u_symbolic)), ctypes.c_double(float(u_runtime)))

    def find_zero_pivot(self, gid):
        return self.groups[gid].zero_slot

    def allreduce_schur(self, comm):
        if comm.size > 1:
            buf = comm.allreduce_sum(np.concatenate([self.S.ravel(), self.tail]))
            self.S = buf[:-8].reshape(self.nc, self.nc)
            self.tail = buf[-8:]

    def factor_schur(self, Q):
        self.Sfull = self.S + (0.0 if Q is None else Q)
        self.Q = None if Q is None else np.array(Q, dtype=np.double)
        n = self.nc
        self.A = np.asfortranarray(np.tril(self.Sfull)).copy(order='F')
        self.ipiv = np.zeros(max(n, 1), dtype=np.int32)
        self.bk = np.zeros(3, dtype=np.int32)
        if n > 0:
            hu.lib().ppsim_bk_factor(n, hu._dp(self.A), hu._ip(self.ipiv), hu._ip(self.bk), ctypes.c_double(1e-14))

    def _flat_to_dense(self, flat):
        gs, G = self.btd
        g2 = gs * gs
        out = np.zeros((gs * G, gs * G))
        for t in range(G):
            out[t * gs:(t + 1) * gs, t * gs:(t + 1) * gs] = flat[t * g2:(t + 1) * g2].reshape(gs, gs).T
        for t in range(G - 1):
            E = flat[(G + t) * g2:(G + t + 1) * g2].reshape(gs, gs).T
            out[(t + 1) * gs:(t + 2) * gs, t * gs:(t + 1) * gs] = E
            out[t * gs:(t + 1) * gs, (t + 1) * gs:(t + 2) * gs] = E.T
        return out

    def factor_schur_flat(self, Qflat):
        self.factor_schur(None if Qflat is None else self._flat_to_dense(np.asarray(Qflat)))

    def factor_schur_corner(self, pos, val):
        gs, G = self.btd
        flat = np.zeros((2 * G - 1) * gs * gs)
        np.add.at(flat, np.asarray(pos, dtype=np.int64), np.asarray(val, dtype=np.float64))
        self.factor_schur_flat(flat)

    def get_schur_flat(self):
        gs, G = self.btd
        g2 = gs * gs
        flat = np.zeros((2 * G - 1) * g2)
        for t in range(G):
            flat[t * g2:(t + 1) * g2] = self.S[t * gs:(t + 1) * gs, t * gs:(t + 1) * gs].T.ravel()
        for t in range(G - 1):
            flat[(G + t) * g2:(G + t + 1) * g2] = self.S[(t + 1) * gs:(t + 2) * gs, t * gs:(t + 1) * gs].T.ravel()
        return flat

    def status(self):
        pos = int(round(self.tail[1])) + int(self.bk[0])
        neg = int(round(self.tail[2])) + int(self.bk[1])
        zero = int(round(self.tail[0])) + int(self.bk[2])
        hs = self.tail[3]
        st = 2 if (zero > 0 or (self.growth_count() > 0 and hu.lib().ppsim_growth_fatal())) else 0
        if hs >= 1e6:
            st = 3
        elif hs >= 1e3:
            st = 2
        elif hs >= 1.0 and st == 0:
            st = 1
        return st, pos, neg, zero

    def get_schur(self):
        return self.S.copy()

    def upload_rhs(self, gid, rhs):
        self.groups[gid].rhs = np.array(rhs, dtype=np.double, copy=True)

    def solve_forward(self, early=False):
        L = hu.lib()
        self.rs = np.zeros(self.nc)
        for sg in self.groups:
            n = sg.g.n
            sg.W = []
            for b in range(sg.batch):
                W = np.zeros(n + sg.m)
                L.ppsim_forward(sg.h, hu._dp(sg.L[b]), hu._dp(np.ascontiguousarray(sg.rhs[b])), hu._dp(W))
                if sg.cmaps is None:
                    self.rs += W[n:]
                else:
                    np.add.at(self.rs, sg.cmaps[b], W[n:])
                sg.W.append(W)

    def allreduce_rs(self, comm):
        if comm.size > 1:
            self.rs = comm.allreduce_sum(self.rs)

    def solve_coupling(self, rc, _refining=False):
        if not _refining and getattr(self, '_refine_saved', None) is None:
            self._last_rc = np.zeros(self.nc) if rc is None else np.array(np.asarray(rc, dtype=np.double)[:self.nc], copy=True)
        b = self.rs + (0.0 if rc is None else np.asarray(rc, dtype=np.double))
        b = np.ascontiguousarray(b, dtype=np.double)
        if self.nc > 0:
            hu.lib().ppsim_bk_solve(self.nc, hu._dp(self.A), hu._ip(self.ipiv), hu._dp(b))
        self.xc = b

    def solve_backward(self):
        L = hu.lib()
        for sg in self.groups:
            n = sg.g.n
            sg.x = np.zeros((sg.batch, n))
            for b in range(sg.batch):
                X = np.zeros(n + sg.m)
                X[n:] = self.xc if sg.cmaps is None else self.xc[sg.cmaps[b]]
                x = np.zeros(n)
                L.ppsim_backward(sg.h, hu._dp(sg.L[b]), hu._dp(sg.Dinv[b]), hu._dp(sg.W[b]), hu._dp(X), hu._dp(x))
                sg.x[b] = x

    def download_solution(self, gid, out):
        out[...] = self.groups[gid].x

    # ---- a-posteriori check and refinement (csrc/refine.hip: pp_residual, pp_refine_begin / _end) ---------------------
    def residual(self, store=False, bc_rhs=None, on_device=False):
        """(rho, group, slot) of the worst local instance: rho = max |b - K x - A^T x_c| / max (|K||x| + |A^T x_c| + |b|)
        over the rows of a block, from the canonical values of the last factorisation; then x_c, sum_i A_i x_i,
        sum_i |A_i||x_i| (library order of the coupling variables) and bc handed back."""
        worst = (0.0, -1, -1)
        scale = 0.0
        ax, aabs = np.zeros(self.nc), np.zeros(self.nc)
        for gid, sg in enumerate(self.groups):
            g = sg.g
            nK = g.rowK.size
            i, j = np.asarray(g.rowK, dtype=np.int64), np.asarray(g.colK, dtype=np.int64)
            off = i != j
            br, bc = np.asarray(g.rowB, dtype=np.int64), np.asarray(g.colB, dtype=np.int64)      # (bc here: border columns)
            if store:
                sg.R = np.zeros((sg.batch, g.n))
            for b in range(sg.batch):
                can, x, rhs = sg.can[b], sg.x[b], np.asarray(sg.rhs[b], dtype=np.double)
                xc = self.xc if sg.cmaps is None else self.xc[sg.cmaps[b]]
                r = np.array(rhs, copy=True)
                s = np.abs(rhs)
                t = can[:nK] * x[j]
                np.subtract.at(r, i, t)
                np.add.at(s, i, np.abs(t))
                t = can[:nK][off] * x[i[off]]
                np.subtract.at(r, j[off], t)
                np.add.at(s, j[off], np.abs(t))
                if br.size:
                    t = can[nK:] * xc[br]
                    np.subtract.at(r, bc, t)
                    np.add.at(s, bc, np.abs(t))
                    t = can[nK:] * x[bc]
                    glob = br if sg.cmaps is None else np.asarray(sg.cmaps[b], dtype=np.int64)[br]
                    np.add.at(ax, glob, t)
                    np.add.at(aabs, glob, np.abs(t))
                if store:
                    sg.R[b] = r
                rm = np.abs(r).max() if r.size else 0.0
                sm = s.max() if s.size else 0.0
                scale = max(scale, float(sm))
                rho = 0.0 if rm == 0.0 else (rm / sm if sm > 0.0 and np.isfinite(rm) else np.inf)
                if rho > worst[0] or worst[1] < 0:
                    worst = (float(rho), gid, b)
        bcv = np.array(self._last_rc, dtype=np.double) if bc_rhs is None else np.array(np.asarray(bc_rhs)[:self.nc], dtype=np.double)
        if on_device and self.nc > 0:
            # (as the library with coupling_on_device: the coupling rows judged here, their residual kept for the correction)
            Q = np.zeros((self.nc, self.nc)) if self.Q is None else self.Q
            rc = bcv - ax - Q.dot(self.xc)
            sc = np.abs(bcv) + aabs + np.abs(Q).dot(np.abs(self.xc))
            rmax = np.abs(rc).max()
            den = max(float(sc.max()), scale)
            rho_c = 0.0 if rmax == 0.0 else (rmax / den if den > 0.0 and np.isfinite(rmax) else np.inf)
            if store:
                self._resid_rc = rc
            return worst + (scale, float(rho_c), None, None, None, None)
        if on_device:
            return worst + (scale, 0.0, None, None, None, None)
        return worst + (scale, None, self.xc.copy(), ax, aabs, bcv)

    def residual_begin(self, store=False, bc_rhs=None, on_device=False):
        self._resid_pending = self.residual(store, bc_rhs, on_device)

    def residual_end(self):
        out, self._resid_pending = self._resid_pending, None
        return out

    def refine_solve_coupling(self):
        self.solve_coupling(self._resid_rc, _refining=True)

    def refine_begin(self):
        self._refine_saved = (self.xc.copy(), [(sg.rhs, sg.x.copy(), getattr(sg, 'rhs_native', None), getattr(sg, 'x_native', None))
                                               for sg in self.groups])
        for sg in self.groups:
            sg.rhs = sg.R
            if hasattr(sg, 'rhs_native'):
                sg.rhs_native = sg.x_native = None

    def refine_end(self):
        xc, saved = self._refine_saved
        self._refine_saved = None
        self.xc = xc + self.xc
        for sg, (rhs, x, rn, xn) in zip(self.groups, saved):
            sg.rhs = rhs
            sg.x = x + sg.x
            if hasattr(sg, 'rhs_native'):
                sg.rhs_native, sg.x_native = rn, xn
                if xn is not None:
                    xn[:, :sg.batch] = sg.x.T

    def coupling_solution(self):
        return self.xc.copy()

    def synchronize(self):
        pass

    def increase_memory_allocation(self, factor):
        self.mem_factor *= float(factor)


class HostSimBoundaryEngine(HostSimEngine):
    """The host interpreter with the HIP engine's host-boundary entry points (stage_upload, stage_upload_verified,
    upload_rhs_rows, download_solution_rows, copy_rows, alloc_pinned) restated in numpy over the addresses the solver
    hands over: the CPU check of the solver's fast paths (which blocks go where, with which pointers)."""

    def __init__(self):
        super().__init__()
        self.calls = {'stage_upload': 0, 'stage_upload_verified': 0, 'upload_rhs_rows': 0, 'download_rows_async': 0,
                      'download_rows_staged': 0, 'copy_rows': 0, 'compared_blocks': 0, 'verified_blocks': 0}
        # the library's view of the staging rows, kept apart from the solver's own flags (_Group.full_rows): a row holds every
        # entry of its block once all runs were staged into it or the solver wrote it whole and sent it; a subset of the runs
        # (declare_constant_entries) may only be staged into such a row
        self._rows_whole = {}

    def _whole_flags(self, gid, staging):
        rec = self._rows_whole.get(gid)
        if rec is None or rec[0] is not staging:
            rec = self._rows_whole[gid] = (staging, np.zeros(staging.shape[0], dtype=bool))
        return rec[1]

    def upload_values_compact(self, gid, compact, row0=0, nrows=None):
        n = compact.shape[0] - row0 if nrows is None else nrows
        self._whole_flags(gid, compact)[row0:row0 + n] = True
        super().upload_values_compact(gid, compact, row0, nrows)

    @staticmethod
    def _view(addr, n, dtype=np.double):
        import ctypes
        if n == 0:
            return np.zeros(0, dtype=dtype)
        ct = ctypes.c_double if dtype == np.double else ctypes.c_int32
        return np.ctypeslib.as_array((ct * n).from_address(int(addr)))

    def alloc_pinned(self, shape):
        return np.zeros(shape, dtype=np.double)

    def bind_native_vectors(self, gid, rhs, x):
        pass

    def _stage_row(self, g, slot, kd, bd):
        row = g.staging[slot]
        for e0, ln, dst in g.runsK:
            row[dst:dst + ln] = kd[e0:e0 + ln]
        for e0, ln, dst in g.runsB:
            row[dst:dst + ln] = bd[e0:e0 + ln]
        self.upload_values_compact(g.gid, g.staging, slot, 1)      # (marks the row whole)

    def stage_upload(self, g, items, full_check=True):
        self.calls['stage_upload'] += 1
        ref = g.raw_refs
        ok = np.zeros(len(items), dtype=bool)
        for i, (slot, (kr, kc, kd, br, bc, bd)) in enumerate(items):
            self.calls['compared_blocks'] += 1
            same = (kd.size == g.nrawK and bd.size == g.nraw - g.nrawK and np.array_equal(kr, ref[0]) and
                    np.array_equal(kc, ref[1]) and np.array_equal(br, ref[2]) and np.array_equal(bc, ref[3]))
            if same:
                self._stage_row(g, slot, kd, bd)
            ok[i] = same
        return ok

    def stage_upload_verified(self, g, slots, kd_ptr, bd_ptr, runs=None):
        # (runs: the subset of the group's runs the library is to look at -- the rest of the row is in g.staging already)
        self.calls['stage_upload_verified'] += 1
        slots = [int(v) for v in slots]
        assert all(a < b for a, b in zip(slots, slots[1:]))
        for slot, kp, bp in zip(slots, kd_ptr, bd_ptr):
            self.calls['verified_blocks'] += 1
            kd, bd = self._view(int(kp), g.nrawK), self._view(int(bp), g.nraw - g.nrawK)
            if runs is None:
                self._stage_row(g, slot, kd, bd)
            else:
                self.calls['variable_entries'] = self.calls.get('variable_entries', 0) + int(runs[0][:, 1].sum() + runs[1][:, 1].sum())
                if not self._whole_flags(g.gid, g.staging)[slot]:
                    raise AssertionError('a subset of the runs staged into row %d of group %d, which does not hold every entry '
                                         'of its block' % (slot, g.gid))
                row = g.staging[slot]
                for e0, ln, dst in runs[0]:
                    row[dst:dst + ln] = kd[e0:e0 + ln]
                for e0, ln, dst in runs[1]:
                    row[dst:dst + ln] = bd[e0:e0 + ln]
                self.upload_values_compact(g.gid, g.staging, slot, 1)

    def stage_upload_end(self):
        self.calls['stage_upload_end'] = self.calls.get('stage_upload_end', 0) + 1

    def copy_rows(self, dst, rows):
        self.calls['copy_rows'] += 1
        for r, v in rows:
            dst[r] = v

    def upload_rhs_rows(self, g, vectors):
        self.calls['upload_rhs_rows'] += 1
        assert len(vectors) == len(g.blocks)
        for i, v in enumerate(vectors):
            g.rhs_staging[i] = v
        self.upload_rhs(g.gid, g.rhs_staging)

    def download_solution_rows(self, g, pinned, out=None):
        self.calls['download_rows_async' if out is None else 'download_rows_staged'] += 1
        self.download_solution(g.gid, pinned)
        if out is not None:
            out[...] = pinned


class NpTensor(np.ndarray):
    """numpy array with the few tensor methods the solver class calls on device containers (CPU tests only)."""

    def is_contiguous(self):
        return bool(self.flags.c_contiguous)

    def cpu(self):
        return self

    def numpy(self):
        return np.asarray(self)

    def numel(self):
        return int(self.size)

    def copy_(self, other):
        self[...] = np.asarray(other)
        return self

    def zero_(self):
        self[...] = 0.0
        return self


def _tensor(shape):
    return np.zeros(shape, dtype=np.double).view(NpTensor)


class HostSimDeviceEngine(HostSimEngine):
    """The host interpreter with the device-container entry points of the HIP engine (value maps and source tensors of a
    DeviceBlockMatrix, [row][instance] right-hand sides and solutions of a DeviceBlockVector, the diagonal-shift fast path,
    the interior-point step kernels) restated in numpy: the CPU check of the producer's host logic -- pattern groups,
    rank distribution, the loop -- including world_size-2 gloo runs."""

    def __init__(self):
        HostSimEngine.__init__(self)
        self._ops = None

    def ip_ops(self):
        if self._ops is None:
            from hostsim_ip_ops import HostSimIpOps
            self._ops = HostSimIpOps(self)
        return self._ops

    def symbolic(self, nc, groups, btd=None, cinv=None):
        stats = HostSimEngine.symbolic(self, nc, groups, btd=btd, cinv=cinv)
        for sg in self.groups:
            sg.vmap = sg.sources = sg.classes = None
            sg.rhs_native = sg.x_native = None
        self._shift = (0.0, 0.0)
        return stats

    def new_tensor(self, shape):
        return _tensor(shape)

    new_tensor_uninitialized = new_tensor

    def set_value_map(self, gid, nsrc, src, coef):
        self.groups[gid].vmap = (int(nsrc), np.asarray(src, dtype=np.int64), np.asarray(coef, dtype=np.double))

    def bind_source_tensor(self, gid, tensor):
        self.groups[gid].sources = tensor

    def set_diagonal_classes(self, gid, cls):
        sg = self.groups[gid]
        g = sg.g
        cls = np.asarray(cls, dtype=np.int8)
        diag = {int(r): k for k, (r, c) in enumerate(zip(g.rowK, g.colK)) if r == c}
        missing = [r for r in np.flatnonzero(cls) if int(r) not in diag]
        if missing:
            raise hu_status_error(3, 'classed row %d has no diagonal entry in the planned pattern' % missing[0])
        sg.classes = [(diag[int(r)], int(cls[r])) for r in np.flatnonzero(cls)]

    def _raw_from_sources(self):
        for sg in self.groups:
            if sg.sources is None or sg.vmap is None:
                continue
            nsrc, src, coef = sg.vmap
            t = np.asarray(sg.sources)
            vals = np.where((src >= 0)[:, None], t[np.maximum(src, 0), :sg.batch], 1.0) * coef[:, None]
            sg.raw = np.ascontiguousarray(vals.T)

    def numeric_local(self):
        dw, dc = self._shift
        if dw == 0.0 and dc == 0.0:
            return HostSimEngine.numeric_local(self)
        # the diagonal shift of the classed rows goes into duplicate raw entries of their diagonal positions: the
        # canonical sum then carries it, as pp_numeric_local_shifted adds it to the gathered diagonal
        saved = []
        for sg in self.groups:
            g = sg.g
            saved.append(sg.raw)
            raw = np.array(sg.raw, copy=True)
            for can, cl in (sg.classes or ()):
                first = g.can_idx[g.can_ptr[can]]
                raw[:, first] += dw if cl == 1 else -dc
            sg.raw = raw
        try:
            HostSimEngine.numeric_local(self)
        finally:
            for sg, raw in zip(self.groups, saved):
                sg.raw = raw

    def numeric_factor_blocks(self):
        self._raw_from_sources()
        HostSimEngine.numeric_factor_blocks(self)

    def numeric_local_shifted(self, delta_w, delta_c):
        self._shift = (float(delta_w), float(delta_c))
        try:
            self.numeric_local()
        finally:
            self._shift = (0.0, 0.0)

    def bind_native_vectors(self, gid, rhs, x):
        sg = self.groups[gid]
        sg.rhs_native, sg.x_native = rhs, x

    def solve_forward(self, early=False):
        for sg in self.groups:
            if sg.rhs_native is not None:
                sg.rhs = np.ascontiguousarray(np.asarray(sg.rhs_native)[:, :sg.batch].T)
        HostSimEngine.solve_forward(self)

    def solve_coupling_dev(self, rc):
        self.solve_coupling(None if rc is None else np.asarray(rc))

    def solve_backward(self):
        HostSimEngine.solve_backward(self)
        for sg in self.groups:
            if sg.x_native is not None:
                sg.x_native[:, :sg.batch] = sg.x.T

    def copy_coupling_solution(self, tensor):
        tensor[...] = self.xc[:tensor.shape[0]]

    def index_tensor(self, idx):
        return np.ascontiguousarray(idx, dtype=np.int64).view(NpTensor)

    def permute(self, idx_t, src, dst, scatter):
        idx = np.asarray(idx_t)
        if scatter:
            dst[...] = 0.0
            np.asarray(dst)[idx] = np.asarray(src)[:idx.size]
        else:
            np.asarray(dst)[:idx.size] = np.asarray(src)[idx]
