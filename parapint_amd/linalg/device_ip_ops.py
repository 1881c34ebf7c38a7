"""Host side of the interior-point step on device-resident iterates (include/parapint_hip.h: pp_ip_*; SURVEY.md section 8
rows f2 / f4).

``HipIpOps`` binds the six entry points of ``csrc/ipstep.hip`` for one solver handle: barrier diagonals and right-hand side
(parapint/interfaces/interface.py:450-465, 496-538), bound-dual steps and fraction to the boundary (:562-588,
algorithms/interior_point.py:655-758), the step (:619-626) and the convergence measures (:174-317) run as kernels on the
solver's stream over [row][instance] tensors; the scalars the loop's control flow needs come back through a pinned mailbox.
Between ranks the per-rank scalars are all-gathered (RCCL through the library or ``torch.distributed``; the reference has
one scalar all-reduce per reduction hidden in PyNumero's MPIBlockVector, and one of the coupling block of the right-hand
side, mpi_sc_ip_interface.py:470-478) and combined in rank order on every rank, so that all ranks take the same decisions.

The producer (``parapint_amd.interfaces.schur_complement.device_sc_ip_interface``) only talks to this object; the CPU tests
substitute a numpy restatement of the same entry points (tests/hostsim_engine.py), the product never does.
"""
import ctypes

import numpy as np

V_HEAD = 8          # scalars in front of the per-coupling-variable sums in a rank's vector (pp_ip_residuals)


class _Prepared(object):
    """Descriptors of the pattern groups (pp_ip_group array) + the tensors they point to (kept alive)."""

    def __init__(self, arr, descs):
        self.arr, self.descs, self.n = arr, descs, len(descs)
        self.delta = [None] * len(descs)


class HipIpOps(object):
    def __init__(self, engine):
        import torch
        from parapint_amd import _native
        self._torch, self._native = torch, _native
        self.eng = engine
        self.lib, self.ns = engine.lib, engine.ns
        self.device = torch.device('cuda', engine.device)

    # ---- buffers
    def from_host(self, a):
        """Host array -> device tensor (float64, or int32 for index data)."""
        a = np.ascontiguousarray(a)
        return self._torch.from_numpy(a).to(self.device)

    def rows_from_instances(self, a):
        """[instance][row] host array -> [row][instance] device tensor (the kernels' layout), transposed on the device."""
        return self._torch.from_numpy(np.ascontiguousarray(a)).to(self.device).t().contiguous()

    def zeros(self, shape):
        return self._torch.zeros(shape, dtype=self._torch.float64, device=self.device)

    def to_host(self, t):
        return t.cpu().numpy()

    # ---- set-up: bounds and initial point of all lanes at once (elementwise, in place; the same operations in the same
    # order as parapint/interfaces/interface.py:389-419 and algorithms/interior_point.py:761-799 apply per scenario)
    def relax_bounds(self, bounds, n, mi, factor):
        """bounds rows: lb (n) | ub (n) | ineq_lb (mi) | ineq_ub (mi), moved outwards by factor * max(1, |bound|)."""
        if factor == 0:
            return
        for r0, r1, sign in ((0, n, -1.0), (n, 2 * n, 1.0), (2 * n, 2 * n + mi, -1.0), (2 * n + mi, 2 * n + 2 * mi, 1.0)):
            b = bounds[r0:r1]
            b.copy_(b + (sign * factor) * self._torch.clamp(b.abs(), min=1.0))

    def process_initial_point(self, W, bounds, n, mi, nb):
        """W rows x | s (pushed inside their bounds) and z_l | z_u | s_l | s_u from row nb on (1, and 0 where the bound is
        infinite); raises ValueError on crossed or equal bounds like the host loop."""
        t = self._torch
        for v0, cnt, b0, z0 in ((0, n, 0, nb), (n, mi, 2 * n, nb + 2 * n)):
            if cnt == 0:
                continue
            x, lo, hi = W[v0:v0 + cnt], bounds[b0:b0 + cnt], bounds[b0 + cnt:b0 + 2 * cnt]
            width = hi - lo
            bad = t.stack(((width < 0).any(), (width == 0).any())).cpu()
            if bool(bad[0]):
                raise ValueError('Lower bounds for variables/inequalities should not be larger than upper bounds.')
            if bool(bad[1]):
                raise ValueError('Variables and inequalities should not have equal lower and upper bounds.')
            has_lb, has_ub = t.isfinite(lo), t.isfinite(hi)
            out = (x >= hi) | (x <= lo)
            x.copy_(t.where(out & has_lb & ~has_ub, lo + 1,
                            t.where(out & has_ub & ~has_lb, hi - 1,
                                    t.where(out & has_lb & has_ub, 0.5 * (lo + hi), x))))
            zl, zu = W[z0:z0 + cnt], W[z0 + cnt:z0 + 2 * cnt]
            zl.copy_(t.where(t.isneginf(lo), 0.0, t.where(zl <= 0, 1.0, zl)))
            zu.copy_(t.where(t.isinf(hi), 0.0, t.where(zu <= 0, 1.0, zu)))

    # ---- descriptors
    def prepare(self, descs):
        arr = (self._native.IpGroup * len(descs))()
        for q, d in zip(arr, descs):
            for k in ('n', 'mi', 'me', 'nfs', 'batch', 'bpad', 'src_dp', 'src_ds'):
                setattr(q, k, int(d[k]))
            q.nfw, q.ncz = int(d.get('nfw', 0)), int(d.get('ncz', 0))
            q.obj_row, q.reserved = int(d.get('obj_row', -1)), 0
            if q.obj_row >= int(d['data'].shape[0]):
                raise ValueError('interior-point step: obj_row is outside the data rows')
            zoff = d.get('zoff')                   # mapped groups (time blocks): [2][bpad] int32 offsets into the coupling states
            if zoff is not None and (not zoff.is_cuda or not zoff.is_contiguous() or zoff.dtype != self._torch.int32 or
                                     tuple(zoff.shape) != (2, int(d['bpad']))):
                raise ValueError('interior-point step: zoff must be a contiguous [2][bpad] int32 device tensor')
            if zoff is not None:
                # the kernels index the coupling vectors with these offsets: checked once, here, on a host copy
                zo = zoff.cpu().numpy()
                if zo.min() < 0 or zo[0].max() + q.nfs > q.ncz or zo[1].max() + q.nfw > q.ncz:
                    raise ValueError('interior-point step: zoff points outside the %d coupling states' % q.ncz)
            q.zoff = None if zoff is None else zoff.data_ptr()
            for k in ('W', 'bounds', 'data', 'src', 'G', 'rhs', 'prog', 'terms'):
                t = d[k]
                if not t.is_contiguous() or not t.is_cuda:
                    raise ValueError('interior-point step: %s must be a contiguous device tensor' % k)
                setattr(q, k, t.data_ptr())
            q.delta = None
        return _Prepared(arr, descs)

    def set_delta(self, hd, gi, t):
        if not t.is_contiguous() or tuple(t.shape) != tuple(hd.descs[gi]['rhs'].shape):
            raise ValueError('interior-point step: the solution of group %d must be a contiguous [n][bpad] tensor' % gi)
        hd.arr[gi].delta = t.data_ptr()
        hd.delta[gi] = t                       # (kept alive until it is replaced)

    # ---- the kernels (all stream-ordered on the solver's stream; only wait() synchronises)
    def rhs(self, hd, mu):
        self.ns.check(self.lib.pp_ip_rhs(self.ns.h, hd.n, hd.arr, float(mu)), 'pp_ip_rhs')

    def step_lengths(self, hd, tau, mu, alpha_local):
        self.ns.check(self.lib.pp_ip_step_lengths(self.ns.h, hd.n, hd.arr, float(tau), float(mu), alpha_local.data_ptr()),
                      'pp_ip_step_lengths')

    def take_step(self, hd, alpha_table, nranks, unified, mu, z, dz):
        self.ns.check(self.lib.pp_ip_take_step(self.ns.h, hd.n, hd.arr, None if alpha_table is None else alpha_table.data_ptr(),
                                               int(nranks), 1 if unified else 0, float(mu), z.data_ptr(),
                                               None if dz is None else dz.data_ptr()), 'pp_ip_take_step')

    def residuals(self, hd, z, v_local):
        self.ns.check(self.lib.pp_ip_residuals(self.ns.h, hd.n, hd.arr, z.data_ptr(), v_local.data_ptr()), 'pp_ip_residuals')

    def publish(self, v_table, alpha_table, nranks, ncoup, dual_from, rhs_coupling):
        self.ns.check(self.lib.pp_ip_publish(self.ns.h, v_table.data_ptr(), None if alpha_table is None else alpha_table.data_ptr(),
                                             int(nranks), int(ncoup), int(dual_from), rhs_coupling.data_ptr()), 'pp_ip_publish')

    def wait(self):
        out = np.zeros(10)
        self.ns.check(self.lib.pp_ip_wait(self.ns.h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))), 'pp_ip_wait')
        return out

    # ---- between ranks
    def allgather(self, comm, local, table):
        """table[r] = local of rank r (device tensors).  One rank: the caller passes table = local."""
        if comm.size == 1 and not getattr(comm, 'always_reduce', False):
            return
        if self.eng._direct_rccl(comm):
            self.ns.check(self.lib.pp_comm_allgather(self.ns.h, local.data_ptr(), table.data_ptr(),
                                                     ctypes.c_int64(local.numel())), 'pp_comm_allgather')
        elif comm.device_collectives:
            comm.allgather_tensor_(table, local)
        else:
            table.copy_(self._torch.from_numpy(comm.allgather(local.cpu().numpy())).reshape(table.shape))
