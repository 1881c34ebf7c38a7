"""Host block matrix whose VALUES come as one flat float64 vector per block (opt-in; the host-side sibling of
``DeviceBlockMatrix``).

The reference's interfaces hand ``do_numeric_factorization`` a block matrix of SciPy COO blocks at every iteration
(parapint/interfaces/interface.py: evaluate_primal_dual_kkt_matrix; consumed by
parapint/linalg/schur_complement/mpi_explicit_schur_complement.py:257-402), index arrays included, although only the values
change.  An interface that keeps its values in flat arrays says so with this container: the pattern is the matrix given to
``do_symbolic_factorization`` (the very object), the values one vector per block -- the entries of ``K_ndx.data`` followed by
those of ``A_ndx.data``, in the order of the pattern's blocks.  With all vectors being the rows of ONE 2-D array (row i = the
i-th block this rank owns, ascending block index) the solver stages a whole pattern group with one library call instead of
walking 2 x N Python objects.
"""
import numpy as np


class HostValueMatrix(object):
    """Parameters
    ----------
    pattern: host block matrix (BlockMatrix / MPIBlockMatrix protocol): sparsity pattern and representative values; the
        object given to ``do_symbolic_factorization`` (or wrapped in a HostValueMatrix given to it).  It must not be
        modified while matrices that refer to it are factorised.
    flat_values: {block index: 1-D float64 array} or a 2-D float64 array [owned blocks][entries] (C order, every owned block
        with the same number of entries); entries of a block: ``K.data`` then ``A.data``.  None: the pattern's own values.
    Q: the coupling block (last, last) of this value set (None: the pattern's).
    """

    def __init__(self, pattern, flat_values=None, Q=None):
        self.pattern = pattern
        self.flat_values = flat_values
        self.Q = Q

    # the BlockMatrix protocol of SURVEY.md 8b, served by the pattern
    @property
    def bshape(self):
        return self.pattern.bshape

    @property
    def shape(self):
        return self.pattern.shape

    def get_block(self, i, j):
        last = self.pattern.bshape[0] - 1
        if self.Q is not None and i == last and j == last:
            return self.Q
        return self.pattern.get_block(i, j)

    def get_row_size(self, i):
        return self.pattern.get_row_size(i)

    def __getattr__(self, name):
        if name == 'rank_ownership':
            return getattr(self.pattern, 'rank_ownership')
        raise AttributeError(name)

    def with_values(self, flat_values, Q=None):
        """Another value set over the same pattern (the next iteration's arrays)."""
        return HostValueMatrix(self.pattern, flat_values, self.Q if Q is None else Q)

    def block_values(self, ndx, row=None):
        """(K data, A data) of one block as views of its flat vector (row: its row in a 2-D array)."""
        last = self.pattern.bshape[0] - 1
        K, A = self.pattern.get_block(ndx, ndx), self.pattern.get_block(last, ndx)
        nK = K.nnz if hasattr(K, 'nnz') else np.asarray(K.data).size
        if self.flat_values is None:
            return np.asarray(K.tocoo().data), (np.zeros(0) if A is None else np.asarray(A.tocoo().data))
        # (a 2-D array is indexed by the block's ROW -- its position among the blocks this rank owns --, every other
        # container -- dict, list of vectors -- by the block index, as the solver's staging does)
        v = self.flat_values[row] if isinstance(self.flat_values, np.ndarray) else self.flat_values[ndx]
        return v[:nK], v[nK:]

    def to_block_matrix(self, local_block_indices=None):
        """An ordinary block matrix with these values (SciPy COO blocks over the pattern's index arrays): what the
        reference's interface would have handed over.  local_block_indices: the blocks the rows of a 2-D ``flat_values``
        belong to, in row order -- the solver passes its own list (blocks with ownership -1 count on rank 0 only,
        mpi_explicit_schur_complement.py:199-203); default: every block the pattern holds."""
        from scipy.sparse import coo_matrix
        pat = self.pattern
        nb = pat.bshape[0]
        last = nb - 1
        out = pat.copy_structure() if hasattr(pat, 'copy_structure') else None
        if out is None:
            from parapint_amd.sparse.block_containers import BlockMatrix
            out = BlockMatrix(nb, nb)
        owned = [i for i in range(last) if pat.get_block(i, i) is not None] if local_block_indices is None \
            else list(local_block_indices)
        for pos, ndx in enumerate(owned):
            K = pat.get_block(ndx, ndx).tocoo()
            A = pat.get_block(last, ndx)
            kd, ad = self.block_values(ndx, pos)
            out.set_block(ndx, ndx, coo_matrix((np.array(kd), (K.row, K.col)), shape=K.shape))
            if A is not None:
                A = A.tocoo()
                Anew = coo_matrix((np.array(ad), (A.row, A.col)), shape=A.shape)
                out.set_block(last, ndx, Anew)
                if pat.get_block(ndx, last) is not None:
                    out.set_block(ndx, last, Anew.transpose().tocoo())
        Q = self.get_block(last, last)
        if Q is not None:
            out.set_block(last, last, Q)
        return out
