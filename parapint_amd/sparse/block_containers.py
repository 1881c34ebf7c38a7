"""Pyomo-free block containers exposing the members the hot path touches.

The reference hands the linear solver PyNumero ``BlockMatrix`` / ``BlockVector`` /
``MPIBlockMatrix`` / ``MPIBlockVector`` objects (not vendored, not installed here).
The solver only uses a small protocol:

  matrix:  .bshape, .shape, .get_block(i, j), .get_row_size(i), .rank_ownership[i, j]
           (reference call sites: mpi_explicit_schur_complement.py:193, 201-202,
           209, 237, 244, 294, 347, 383; explicit_schur_complement.py:60-65)
  vector:  .get_block(i), .set_block(i, v), .copy_structure(), .flatten(),
           .copyfrom(), .nblocks (mpi_...:381-398; scipy_interface.py:50-62)

These classes implement exactly that protocol (plus ``tocoo``/``toarray`` so the
synthetic generator and tests can assemble full-space systems).  Real PyNumero
containers are accepted by the solver unchanged because only the protocol is used.
"""
import numpy as np
from scipy.sparse import coo_matrix, isspmatrix


def _is_block(obj):
    return isinstance(obj, (BlockMatrix,))


# flat index arrays of the nested structures seen so far: signature -> [(leaf index arrays, flat rows, flat cols)]
_STRUCTURES = {}
_STRUCTURES_MAX = 256


class BlockMatrix(object):
    """2-D grid of sparse blocks; empty blocks are structural zeros."""

    def __init__(self, nbrows, nbcols):
        self._nbrows = int(nbrows)
        self._nbcols = int(nbcols)
        self._blocks = {}
        self._row_sizes = [None] * self._nbrows
        self._col_sizes = [None] * self._nbcols

    # -- structure -----------------------------------------------------
    @property
    def bshape(self):
        return (self._nbrows, self._nbcols)

    @property
    def shape(self):
        self._require_sizes()
        return (int(sum(self._row_sizes)), int(sum(self._col_sizes)))

    def _require_sizes(self):
        if any(s is None for s in self._row_sizes) or any(s is None for s in self._col_sizes):
            raise RuntimeError('BlockMatrix has block rows/columns of undefined size')

    def set_row_size(self, i, size):
        self._row_sizes[i] = int(size)

    def set_col_size(self, j, size):
        self._col_sizes[j] = int(size)

    def get_row_size(self, i):
        if self._row_sizes[i] is None:
            raise RuntimeError('row size %d undefined' % i)
        return self._row_sizes[i]

    def get_col_size(self, j):
        if self._col_sizes[j] is None:
            raise RuntimeError('col size %d undefined' % j)
        return self._col_sizes[j]

    def row_block_sizes(self):
        self._require_sizes()
        return np.asarray(self._row_sizes, dtype=np.int64)

    def col_block_sizes(self):
        self._require_sizes()
        return np.asarray(self._col_sizes, dtype=np.int64)

    def is_empty_block(self, i, j):
        return (i, j) not in self._blocks

    # -- blocks --------------------------------------------------------
    def set_block(self, i, j, block):
        if block is None:
            self._blocks.pop((i, j), None)
            return
        if not (isspmatrix(block) or isinstance(block, BlockMatrix)):
            block = coo_matrix(np.atleast_2d(np.asarray(block, dtype=np.double)))
        nr, nc = block.shape
        if self._row_sizes[i] is None:
            self._row_sizes[i] = int(nr)
        elif self._row_sizes[i] != nr:
            raise ValueError('block row size mismatch at (%d, %d)' % (i, j))
        if self._col_sizes[j] is None:
            self._col_sizes[j] = int(nc)
        elif self._col_sizes[j] != nc:
            raise ValueError('block col size mismatch at (%d, %d)' % (i, j))
        self._blocks[(i, j)] = block

    def get_block(self, i, j):
        return self._blocks.get((i, j), None)

    # -- conversions ---------------------------------------------------
    def _leaves(self, r0, c0, out):
        """(COO leaf, row offset, column offset) of every stored block, nested ones flattened, in storage order."""
        self._require_sizes()
        roff = np.concatenate([[0], np.cumsum(self._row_sizes)])
        coff = np.concatenate([[0], np.cumsum(self._col_sizes)])
        for (i, j), blk in self._blocks.items():
            if isinstance(blk, BlockMatrix):
                blk._leaves(r0 + int(roff[i]), c0 + int(coff[j]), out)
            else:
                out.append((blk if getattr(blk, 'format', None) == 'coo' else blk.tocoo(), r0 + int(roff[i]), c0 + int(coff[j])))
        return out

    def tocoo(self, copy_index=False):
        """Flat COO matrix.  The interior-point interfaces build a NEW nested matrix at every iteration with the same
        structure (interface.py:432-494, sc_ip_interface.py:839-843), and the solver flattens one per block: the index
        arrays of a structure seen before are not built again -- the leaves' index arrays are compared with remembered
        COPIES (by contents: a leaf array rewritten in place is therefore seen) and the SAME flat index arrays are handed out, so that only the values are
        concatenated (1.5 -> 0.15 ms for a 9200-row KKT block) and a caller that recognises index arrays by address (the
        HIP solver's staging) sees one pattern object for all blocks and iterations.  The returned index arrays are shared
        between all matrices of that structure and are therefore READ-ONLY (numpy raises on an in-place change such as
        ``coo.row += offset`` or ``coo.sum_duplicates()``; take a copy first); they are int32 while the dimension allows it
        (widen before forming ``row * n + col``).  ``tocoo(copy_index=True)``: private, writable index arrays -- what
        PyNumero's BlockMatrix.tocoo() returns -- for callers that change them in place."""
        if copy_index:
            shared = self.tocoo()
            return coo_matrix((shared.data, (shared.row.copy(), shared.col.copy())), shape=shared.shape, copy=False)
        leaves = self._leaves(0, 0, [])
        shape = self.shape
        sig = (shape,) + tuple((r, c, lf.nnz) + lf.shape for lf, r, c in leaves)
        for entry in _STRUCTURES.get(sig, ()):
            if all(np.array_equal(lf.row, kr) and np.array_equal(lf.col, kc)
                   for (lf, _, _), (kr, kc) in zip(leaves, entry[0])):
                data = np.concatenate([np.asarray(lf.data, dtype=np.double) for lf, _, _ in leaves]) if leaves \
                    else np.zeros(0, dtype=np.double)
                return coo_matrix((data, (entry[1], entry[2])), shape=shape, copy=False)
        idt = np.int32 if max(shape) < 2 ** 31 else np.int64
        if leaves:
            rows = np.concatenate([lf.row.astype(idt) + idt(r) for lf, r, _ in leaves])
            cols = np.concatenate([lf.col.astype(idt) + idt(c) for lf, _, c in leaves])
            data = np.concatenate([np.asarray(lf.data, dtype=np.double) for lf, _, _ in leaves])
        else:
            rows, cols, data = np.zeros(0, dtype=idt), np.zeros(0, dtype=idt), np.zeros(0, dtype=np.double)
        if len(_STRUCTURES) >= _STRUCTURES_MAX:
            _STRUCTURES.pop(next(iter(_STRUCTURES)))                # (oldest signature first)
        known = _STRUCTURES.setdefault(sig, [])
        if len(known) >= 8:                                         # (patterns that differ only in their indices: keep the last few)
            known.pop(0)
        rows.setflags(write=False)
        cols.setflags(write=False)
        known.append(([(lf.row.copy(), lf.col.copy()) for lf, _, _ in leaves], rows, cols))
        return coo_matrix((data, (rows, cols)), shape=shape, copy=False)

    def tocsr(self):
        return self.tocoo().tocsr()

    def tocsc(self):
        return self.tocoo().tocsc()

    def toarray(self):
        return self.tocoo().toarray()

    def transpose(self, copy=True):
        res = BlockMatrix(self._nbcols, self._nbrows)
        res._row_sizes = list(self._col_sizes)
        res._col_sizes = list(self._row_sizes)
        for (i, j), blk in self._blocks.items():
            res._blocks[(j, i)] = blk.transpose(copy=copy) if not isinstance(blk, BlockMatrix) else blk.transpose(copy)
        return res

    def copy_structure(self):
        res = BlockMatrix(self._nbrows, self._nbcols)
        res._row_sizes = list(self._row_sizes)
        res._col_sizes = list(self._col_sizes)
        return res

    def copy(self):
        """Deep copy (nested BlockMatrix blocks included), as PyNumero's BlockMatrix.copy()."""
        res = self.copy_structure()
        for k, blk in self._blocks.items():
            res._blocks[k] = blk.copy()
        return res

    def __add__(self, other):
        if isinstance(other, BlockMatrix):
            assert other.bshape == self.bshape
            res = self.copy_structure()
            keys = set(self._blocks) | set(other._blocks)
            for k in keys:
                a = self._blocks.get(k)
                b = other._blocks.get(k)
                if a is None:
                    res.set_block(k[0], k[1], b.copy())
                elif b is None:
                    res.set_block(k[0], k[1], a.copy())
                else:
                    res.set_block(k[0], k[1], (a.tocoo() + b.tocoo()).tocoo())
            return res
        return self.tocoo() + other

    def __mul__(self, other):
        if isinstance(other, BlockVector):
            other = other.flatten()
        return self.tocsr() * other

    dot = __mul__


class BlockVector(object):
    """1-D list of vector blocks (ndarray or nested BlockVector)."""

    def __init__(self, nblocks):
        self._nblocks = int(nblocks)
        self._blocks = [None] * self._nblocks

    @property
    def nblocks(self):
        return self._nblocks

    @property
    def bshape(self):
        return (self._nblocks,)

    @property
    def size(self):
        return int(sum(b.size for b in self._blocks if b is not None))

    @property
    def shape(self):
        return (self.size,)

    def set_block(self, i, v):
        if not isinstance(v, BlockVector):
            v = np.asarray(v, dtype=np.double)
        self._blocks[i] = v

    def get_block(self, i):
        return self._blocks[i]

    def flatten(self):
        parts = []
        for b in self._blocks:
            if b is None:
                raise RuntimeError('BlockVector has undefined blocks')
            parts.append(b.flatten() if isinstance(b, BlockVector) else np.asarray(b, dtype=np.double).ravel())
        if not parts:
            return np.zeros(0, dtype=np.double)
        return np.concatenate(parts)

    def copy_structure(self):
        res = BlockVector(self._nblocks)
        for i, b in enumerate(self._blocks):
            if b is None:
                continue
            if isinstance(b, BlockVector):
                res._blocks[i] = b.copy_structure()
            else:
                res._blocks[i] = np.zeros(b.size, dtype=np.double)
        return res

    def copy_structure_unset(self):
        """Same block layout with every block unset: for callers that set_block() each block they own anyway
        (zero-filling 75 MB of blocks per back-solve costs 11 ms at 1024 x 9200).  Not a PyNumero method."""
        return BlockVector(self._nblocks)

    def copyfrom(self, other):
        if isinstance(other, BlockVector):
            other = other.flatten()
        other = np.asarray(other, dtype=np.double)
        off = 0
        for i, b in enumerate(self._blocks):
            if b is None:
                raise RuntimeError('BlockVector has undefined blocks')
            n = b.size
            if isinstance(b, BlockVector):
                b.copyfrom(other[off:off + n])
            else:
                self._blocks[i] = other[off:off + n].copy()
            off += n
        if off != other.size:
            raise ValueError('size mismatch in copyfrom')

    def copy(self):
        res = BlockVector(self._nblocks)
        for i, b in enumerate(self._blocks):
            if b is not None:
                res._blocks[i] = b.copy()
        return res

    def __sub__(self, other):
        res = self.copy_structure()
        res.copyfrom(self.flatten() - (other.flatten() if isinstance(other, BlockVector) else other))
        return res

    def __add__(self, other):
        res = self.copy_structure()
        res.copyfrom(self.flatten() + (other.flatten() if isinstance(other, BlockVector) else other))
        return res

    def __array__(self, dtype=None, copy=None):
        return self.flatten()


class MPIBlockMatrix(BlockMatrix):
    """BlockMatrix with a rank-ownership table (-1 = owned by every rank).

    Only ``rank_ownership`` and the BlockMatrix protocol are used by the solver
    (reference: mpi_explicit_schur_complement.py:199-203).  ``mpi_comm`` is any
    object with ``rank``/``size`` attributes (see parapint_amd/linalg/comm.py).
    """

    def __init__(self, nbrows, nbcols, rank_ownership, mpi_comm=None, assert_correct_owners=False):
        super().__init__(nbrows, nbcols)
        self._rank_owner = np.asarray(rank_ownership, dtype=np.int64)
        assert self._rank_owner.shape == (nbrows, nbcols)
        self._mpiw = mpi_comm

    @property
    def rank_ownership(self):
        return self._rank_owner

    @property
    def mpi_comm(self):
        return self._mpiw

    def broadcast_block_sizes(self):
        """Make block sizes known on every rank (all-reduce MAX of the size tables)."""
        comm = self._mpiw
        rs = np.array([-1 if s is None else s for s in self._row_sizes], dtype=np.int64)
        cs = np.array([-1 if s is None else s for s in self._col_sizes], dtype=np.int64)
        if comm is not None and comm.size > 1:
            rs = comm.allreduce_max_int(rs)
            cs = comm.allreduce_max_int(cs)
        self._row_sizes = [None if s < 0 else int(s) for s in rs]
        self._col_sizes = [None if s < 0 else int(s) for s in cs]

    def copy_structure(self):
        res = MPIBlockMatrix(self._nbrows, self._nbcols, self._rank_owner, self._mpiw)
        res._row_sizes = list(self._row_sizes)
        res._col_sizes = list(self._col_sizes)
        return res


class MPIBlockVector(BlockVector):
    """BlockVector with a rank-owner list (-1 = replicated on every rank)."""

    def __init__(self, nblocks, rank_owner, mpi_comm=None, assert_correct_owners=False):
        super().__init__(nblocks)
        self._rank_owner = np.asarray(rank_owner, dtype=np.int64)
        assert self._rank_owner.shape == (nblocks,)
        self._mpiw = mpi_comm
        self._block_sizes = [None] * nblocks

    @property
    def rank_ownership(self):
        return self._rank_owner

    @property
    def mpi_comm(self):
        return self._mpiw

    def set_block(self, i, v):
        super().set_block(i, v)
        self._block_sizes[i] = self._blocks[i].size

    def owned_blocks(self):
        rank = 0 if self._mpiw is None else self._mpiw.rank
        return [i for i in range(self._nblocks) if self._rank_owner[i] in (rank, -1)]

    def broadcast_block_sizes(self):
        comm = self._mpiw
        bs = np.array([-1 if s is None else s for s in self._block_sizes], dtype=np.int64)
        if comm is not None and comm.size > 1:
            bs = comm.allreduce_max_int(bs)
        self._block_sizes = [None if s < 0 else int(s) for s in bs]

    def copy_structure_unset(self):
        res = MPIBlockVector(self._nblocks, self._rank_owner, self._mpiw)
        res._block_sizes = list(self._block_sizes)
        return res

    def copy_structure(self):
        res = MPIBlockVector(self._nblocks, self._rank_owner, self._mpiw)
        res._block_sizes = list(self._block_sizes)
        for i, b in enumerate(self._blocks):
            if b is None:
                continue
            if isinstance(b, BlockVector):
                res._blocks[i] = b.copy_structure()
            else:
                res._blocks[i] = np.zeros(b.size, dtype=np.double)
        return res

    def make_local_copy(self):
        """Gather every block onto every rank and return a plain BlockVector."""
        comm = self._mpiw
        res = BlockVector(self._nblocks)
        rank = 0 if comm is None else comm.rank
        for i in range(self._nblocks):
            owner = int(self._rank_owner[i])
            if owner == -1 or comm is None or comm.size == 1:
                res.set_block(i, np.asarray(self._blocks[i], dtype=np.double).copy()
                              if not isinstance(self._blocks[i], BlockVector) else self._blocks[i].flatten())
                continue
            n = self._block_sizes[i]
            if n is None:
                raise RuntimeError('call broadcast_block_sizes() before make_local_copy()')
            buf = np.zeros(n, dtype=np.double)
            if owner == rank:
                b = self._blocks[i]
                buf[:] = b.flatten() if isinstance(b, BlockVector) else b
            buf = comm.allreduce_sum(buf)
            res.set_block(i, buf)
        return res
