"""Helpers of the HIP Schur-complement solver class (parapint_amd.linalg.hip_schur_complement): timer labels, COO access
to the caller's blocks, canonical (lower-triangular, duplicate-free) patterns, the per-group host state."""
import numpy as np

from parapint_amd.linalg.results import LinearSolverStatus

class _NullTimer(object):
    def start(self, name):
        pass

    def stop(self, name):
        pass


_ROCTX = None


def _roctx():
    """roctx range functions (rocprofv3 --marker-trace shows the reference's timer labels as ranges), or False."""
    global _ROCTX
    if _ROCTX is None:
        _ROCTX = False
        import os
        if os.environ.get('PP_ROCTX', '0') not in ('', '0'):
            import ctypes
            for name in ('librocprofiler-sdk-roctx.so', 'libroctx64.so'):
                try:
                    lib = ctypes.CDLL(name)
                    lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                    _ROCTX = (lib.roctxRangePushA, lib.roctxRangePop)
                    break
                except (OSError, AttributeError):
                    continue
    return _ROCTX


class _Labels(object):
    """The reference's HierarchicalTimer labels (mpi_...:207-255, 291-360), mirrored as roctx ranges when PP_ROCTX=1."""

    def __init__(self, timer):
        self._t = _NullTimer() if timer is None else timer
        self._r = _roctx()

    def start(self, name):
        self._t.start(name)
        if self._r:
            self._r[0](name.encode())

    def stop(self, name):
        if self._r:
            self._r[1]()
        self._t.stop(name)


from parapint_amd.linalg.hip_engine import (HipEngine, _S8, _addr, _checksum, _index_intact,  # noqa: F401 (HipEngine: re-exported)
                                            _index_record)


def _flat(v):
    return v.flatten() if hasattr(v, 'get_block') else np.asarray(v, dtype=np.double).ravel()


def _coo(block):
    """(row, col, data) of a SciPy sparse matrix or (nested) BlockMatrix block."""
    if getattr(block, 'format', None) == 'coo':          # (1024 blocks per call: skip tocoo() / asarray() when there is nothing to do)
        d = block.data
        return block.row, block.col, d if d.dtype == np.double else d.astype(np.double), block.shape
    c = block.tocoo()
    return c.row, c.col, np.asarray(c.data, dtype=np.double), c.shape


def _canonical(row, col, ncols, lower_only):
    """Unique (sorted) pattern of the entries kept, and the CSR map canonical -> raw indices."""
    if lower_only:
        idx = np.flatnonzero(row >= col)
        key = col[idx].astype(np.int64) * ncols + row[idx]        # column-major order of tril
    else:
        idx = np.arange(row.size)
        key = row.astype(np.int64) * ncols + col
    order = np.argsort(key, kind='stable')
    ks = key[order]
    first = np.ones(ks.size, dtype=bool)
    first[1:] = ks[1:] != ks[:-1]
    starts = np.flatnonzero(first)
    can_ptr = np.concatenate([starts, [ks.size]]).astype(np.int32)
    can_idx = idx[order].astype(np.int32)
    uk = ks[first]
    if lower_only:
        crow, ccol = (uk % ncols).astype(np.int32), (uk // ncols).astype(np.int32)
    else:
        crow, ccol = (uk // ncols).astype(np.int32), (uk % ncols).astype(np.int32)
    return crow, ccol, can_ptr, can_idx


class _PatternChanged(Exception):
    """A block carries entries outside the pattern the plan was made for (e.g. the diagonal blocks the
    inertia-correction loop adds, interior_point.py:377-378 / sc_ip_interface.py:1736-1757)."""


class _UnionMatrix(object):
    """Minimal block-matrix view (the protocol of SURVEY 8b) over per-block COO matrices; used to re-plan on
    the union of the old and the new pattern."""

    def __init__(self, nb, blocks, nc):
        self.bshape = (nb, nb)
        self._blocks = blocks
        self._nc = nc

    def get_block(self, i, j):
        return self._blocks.get((i, j))

    def get_row_size(self, i):
        return self._nc if i == self.bshape[0] - 1 else self._blocks[(i, i)].shape[0]


class _BlockInfo(object):
    __slots__ = ('group', 'slot', 'raw_sig', 'n', 'cmap', 'br_cache', 'seen')


class _Group(object):
    """Host-side description of one pattern group (blocks sharing tril(K_i) and A_i patterns)."""

    def __init__(self, n, rowK, colK, rowB, colB, can_ptr, can_idx, nrawK, nraw, raw_refs):
        self.n = n
        self.rowK, self.colK, self.rowB, self.colB = rowK, colK, rowB, colB
        self.can_ptr, self.can_idx = can_ptr, can_idx
        self.nrawK, self.nraw = nrawK, nraw
        self.raw_refs = raw_refs            # (rowK_raw, colK_raw, rowB_raw, colB_raw) of the reference block
        self.blocks = []                    # block indices, slot order
        self.rep_vals = None
        self._alloc = None                  # (shape, pinned if possible) -> zeroed host array; set by the solver
        self._staging = self._rhs_staging = self._x_pool = None     # host boundary buffers, allocated at first use
        self.result_buffers = 0
        self.x_shape = None
        self.alt_layouts = []               # other raw COO layouts seen: (kr, kc, br, bc, canonical position per entry)
        self._keyK = None
        self._keyB = None
        # compact staging: only the raw entries some canonical entry reads are staged and uploaded
        self.used = np.unique(can_idx).astype(np.int64) if can_idx.size else np.zeros(0, dtype=np.int64)
        cpos = -np.ones(max(nraw, 1), dtype=np.int64)
        cpos[self.used] = np.arange(self.used.size)
        self.can_cidx = cpos[can_idx] if can_idx.size else np.zeros(0, dtype=np.int64)   # compact position of every raw duplicate
        self.runsK, self.runsB = self._runs(self.used, nrawK)
        self.known_ptrs = {}                # id(index array) -> array: verified equal to the reference arrays (kept alive)
        # set by the solver class (all state a group can carry is named here)
        self.gid = -1                       # index of the group in the library
        self.m = 0                          # coupling rows of a block of this group (local rows for mapped groups)
        self.cmaps = []                     # per block: local -> global coupling rows (mapped groups), else None
        self.x_turn, self.x_pinned = 0, None                       # result buffers of the host boundary
        self.device_sources = None          # [nsrc][padded batch] tensor the factorisation reads its values from (f2)
        self.refresh_futile = 0             # pivot-order refreshes in a row that did not cure a breakdown
        self.refresh_skip = 0               # breakdowns still to be reported `singular` at once (opt-in back-off)
        self.futile_vals = None             # representative values of the last futile refresh
        self._ref32 = self._refptr = None   # int32 copies of raw_refs and their addresses (stage_upload)
        self.var_runs = None                # (runsK, runsB) over the entries not declared constant (declare_constant_entries)
        self.const_src = self.const_dst = None     # raw entry / compact position of the read entries that ARE declared constant
        self.const_parts = None
        self.full_rows = None               # per slot: the staging row holds every entry of its block (made with the staging array)

    # Buffers of the HOST boundary (page-locked when the engine can: ~12 ms per allocation).  A caller that keeps values,
    # right-hand sides and solutions on the device (rows f2/f4) never touches them, so they are made at first use.
    @property
    def staging(self):
        """Compact rows: only the entries that are read."""
        if self._staging is None:
            self._staging = self._alloc((len(self.blocks), self.used.size))
            self.full_rows = np.zeros(len(self.blocks), dtype=bool)      # rows that hold EVERY entry of their block
        return self._staging

    @property
    def rhs_staging(self):
        if self._rhs_staging is None:
            self._rhs_staging = self._alloc((len(self.blocks), self.n))
        return self._rhs_staging

    @property
    def x_pool(self):
        if self._x_pool is None:
            self._x_pool = [self._alloc(self.x_shape, pinned_only=True) for _ in range(self.result_buffers)]
            if any(a is None for a in self._x_pool):
                self._x_pool = []
        return self._x_pool

    @staticmethod
    def _runs(used, nrawK, select=None, gap=0):
        """Maximal runs of consecutive used raw entries as {source start, length, destination} triples, split at the
        boundary between the K data and the border data (include/parapint_hip.h: pp_stage_values_runs).  select (bool per
        used entry): runs over the selected entries only -- destinations stay positions in the whole compact row; two
        selected entries at most `gap` unselected USED entries apart stay in one run (declare_constant_entries)."""
        runsK, runsB = [], []
        if used.size:
            pos = np.arange(used.size) if select is None else np.flatnonzero(select)
            if pos.size:
                raw = used[pos]
                draw, dpos = np.diff(raw), np.diff(pos)
                brk = np.flatnonzero((draw != dpos) | (dpos > gap + 1)) + 1
                starts = np.concatenate([[0], brk])
                ends = np.concatenate([brk, [pos.size]])
                for a, b in zip(starts, ends):
                    e0, e1, d = int(raw[a]), int(raw[b - 1]) + 1, int(pos[a])
                    if e0 < nrawK < e1:                       # a run across the boundary
                        runsK.append((e0, nrawK - e0, d))
                        runsB.append((0, e1 - nrawK, d + nrawK - e0))
                    elif e0 < nrawK:
                        runsK.append((e0, e1 - e0, d))
                    else:
                        runsB.append((e0 - nrawK, e1 - e0, d))
        return (np.asarray(runsK, dtype=np.int64).reshape(-1, 3), np.asarray(runsB, dtype=np.int64).reshape(-1, 3))

    def variable_runs(self, constant, gap=16):
        """Runs over the used raw entries outside `constant` (bool per raw entry, K data then border data), or None when
        nothing that is read is constant.  Also remembers which read entries are constant (raw entry, compact position)."""
        keep = ~np.asarray(constant, dtype=bool)[self.used]
        if keep.all():
            self.const_src = self.const_dst = None
            return None
        self.const_dst = np.flatnonzero(~keep)
        self.const_src = self.used[self.const_dst]
        # (split once into the K part and the border part: the per-call comparison of sampled blocks indexes with these)
        k_part = self.const_src < self.nrawK
        self.const_parts = (self.const_src[k_part], self.const_dst[k_part], self.const_src[~k_part] - self.nrawK, self.const_dst[~k_part])
        return self._runs(self.used, self.nrawK, select=keep, gap=gap)

    def canonical_from_compact(self, row):
        """Canonical values (duplicates summed) from one compact staging row."""
        return np.add.reduceat(row[self.can_cidx], self.can_ptr[:-1]) if self.can_cidx.size else np.zeros(0)

    def keys(self):
        """Sorted int64 keys of the canonical K (column-major tril) and border (row-major) patterns."""
        if self._keyK is None:
            self._keyK = self.colK.astype(np.int64) * self.n + self.rowK
            self._keyB = self.rowB.astype(np.int64) * self.n + self.colB
        return self._keyK, self._keyB


# Severity of a status when ranks disagree: the reference lets the first failing status win (mpi_...:19-30,
# explicit_...:9-13); with one reduction the worst one wins, `warning` being the mildest non-success.
_SEVERITY = {LinearSolverStatus.successful: 0, LinearSolverStatus.warning: 1, LinearSolverStatus.not_enough_memory: 2,
             LinearSolverStatus.singular: 3, LinearSolverStatus.error: 4}
_BY_SEVERITY = {v: k for k, v in _SEVERITY.items()}
