"""Device-resident counterparts of the block containers.

``DeviceBlockMatrix`` is what an interior-point interface hands to ``do_symbolic_factorization`` /
``do_numeric_factorization`` when it keeps its per-iteration arrays in HBM: the *pattern* is an ordinary host block
matrix (fixed after the symbolic phase, reference contract interior_point.py:542), the *values* are one device array
of "sources" per pattern group -- the interface's own Hessian / Jacobian / barrier-diagonal arrays
(interfaces/interface.py:432-494) in [source][instance] order -- and a value map tells which source (times which
coefficient) every COO entry of K_i and A_i is.  The solver gathers its input from the sources
(include/parapint_hip.h: pp_set_value_map); nothing is assembled, staged or uploaded per iteration.

``DeviceBlockVector`` holds the right-hand sides / solutions of the local blocks as one [n][padded batch] device array
per pattern group (the kernels' own layout: no transposition on either side of the solve) plus the coupling block, so that the step after the solve (convergence check, fraction to the
boundary, interior_point.py:174-317, 655-758) can run on the device (parapint_amd.linalg.device_vector_ops).
"""
import numpy as np


class DeviceBlockMatrix(object):
    """Parameters
    ----------
    pattern: host block matrix (BlockMatrix / MPIBlockMatrix protocol) with the sparsity pattern and representative
        values (they fix the static pivot order, as the matrix given to the reference's symbolic phase does)
    value_maps: {block index: (src, coef)} -- for the COO entries of K_ndx followed by those of A_ndx (the order of
        ``get_block(ndx, ndx).tocoo()`` / ``get_block(last, ndx).tocoo()``): value = coef * source[src], src < 0: the
        constant coef.  Blocks of one pattern group must share one map.
    nsrc: number of source rows per block
    """

    def __init__(self, pattern, value_maps, nsrc):
        self.pattern = pattern
        self.value_maps = value_maps
        self.nsrc = int(nsrc)
        self.sources = {}          # group id -> torch tensor [nsrc][bpad] (float64, device), set by the solver / producer
        self.slots = {}            # group id -> block indices in lane order (instance b of the group = slots[gid][b])
        self.Q = None              # dense host coupling block (or None = zero)
        self.base = None           # with_diagonal_shift: the matrix the shift is relative to
        self.diagonal_shift = None # ... and (delta_w, delta_c, coupling_shift)
        self.coupling_classes = None   # int8 per coupling row: 1 gets +coupling_shift, 2 gets -delta_c (None: all 1)

    # the BlockMatrix protocol of SURVEY.md 8b, served by the pattern
    @property
    def bshape(self):
        return self.pattern.bshape

    @property
    def shape(self):
        return self.pattern.shape

    def get_block(self, i, j):
        return self.pattern.get_block(i, j)

    def get_row_size(self, i):
        return self.pattern.get_row_size(i)

    def __getattr__(self, name):
        if name == 'rank_ownership':
            return getattr(self.pattern, 'rank_ownership')
        raise AttributeError(name)

    def with_sources(self, sources):
        """Another value set of the same matrix structure (e.g. the next iteration's arrays)."""
        other = DeviceBlockMatrix(self.pattern, self.value_maps, self.nsrc)
        other.slots = self.slots
        other.sources = dict(sources)
        other.Q = self.Q
        other.coupling_classes = self.coupling_classes
        return other

    def with_diagonal_shift(self, delta_w=0.0, delta_c=0.0, coupling_shift=0.0):
        """The same matrix + delta_w on the Hessian diagonals - delta_c on the constraint diagonals (the rows classed by
        ``solver.set_regularization_classes``) + coupling_shift * I on the coupling block (with ``coupling_classes``:
        + coupling_shift on its rows of class 1, - delta_c on those of class 2 -- the multipliers of the forward links of a
        time-staged problem live there, sc_ip_interface.py:903-933): what the inertia-correction
        loop builds with ``regularize_hessian`` / ``regularize_equality_gradient`` (interior_point.py:377-386,
        interfaces/interface.py:590-619).  ``do_numeric_factorization`` recognises it and factorises from the values that
        are already on the device."""
        base = self.base if self.diagonal_shift is not None else self
        other = DeviceBlockMatrix(base.pattern, base.value_maps, base.nsrc)
        other.slots, other.sources, other.Q = base.slots, base.sources, base.Q
        other.coupling_classes = base.coupling_classes
        other.base = base
        other.diagonal_shift = (float(delta_w), float(delta_c), float(coupling_shift))
        return other

    def copy(self):
        """(the inertia-correction loop copies the matrix before it regularises it, interior_point.py:383-384)"""
        return self.with_diagonal_shift(*(self.diagonal_shift or (0.0, 0.0, 0.0)))

    def set_sources_from_host(self, per_block):
        """per_block: {block index: source vector (nsrc)} -> the group tensors (test / set-up helper)."""
        import torch
        for gid, blocks in self.slots.items():
            t = self.sources[gid]
            host = np.zeros((self.nsrc, t.shape[1]))
            for b, ndx in enumerate(blocks):
                host[:, b] = per_block[ndx]
            t.copy_(torch.from_numpy(host))


class DeviceBlockVector(object):
    """Right-hand side / solution on the device in the layout the kernels work in: ``group_tensors[gid]`` is
    [n][padded batch] (row = row of the block, column b = block slots[gid][b]; include/parapint_hip.h:
    pp_bind_native_vectors), ``coupling`` is [n_c].  ``get_block(i)`` is a (strided) view of one block."""

    def __init__(self, nblocks, layout):
        self._nblocks = int(nblocks)
        self.layout = layout            # {block index: (gid, slot)}
        self.group_tensors = {}
        self.coupling = None

    @property
    def nblocks(self):
        return self._nblocks

    def get_block(self, i):
        if i == self._nblocks - 1:
            return self.coupling
        gid, slot = self.layout[i]
        return self.group_tensors[gid][:, slot]

    def to_host(self, template):
        """Copy into a host block vector with the structure of `template` (a BlockVector / MPIBlockVector)."""
        out = template.copy_structure_unset() if hasattr(template, 'copy_structure_unset') else template.copy_structure()
        host = {gid: t.cpu().numpy() for gid, t in self.group_tensors.items()}
        for ndx, (gid, slot) in self.layout.items():
            blk = template.get_block(ndx)
            x = np.ascontiguousarray(host[gid][:, slot])
            if hasattr(blk, 'get_block'):
                sub = blk.copy_structure()
                sub.copyfrom(x)
                x = sub
            out.set_block(ndx, x)
        out.set_block(self._nblocks - 1, self.coupling.cpu().numpy())
        return out
