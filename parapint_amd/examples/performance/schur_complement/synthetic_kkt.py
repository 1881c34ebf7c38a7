"""Synthetic block-bordered KKT generator for the Schur-complement perf harness.

Restates the data model of the reference's
``parapint/examples/performance/schur_complement/create_model.py:11-143`` and
``utils.py:6-31`` (parameter-estimation KKT; rows y, q, lambda, nu of sizes
n_y, n_q, n_y, n_theta):

        K_i = [[2I   0    I    0 ],      border  A_i = [0 0 0 -I]   Q = 0
               [0    0  -A^T  P^T],
               [I   -A    0    0 ],
               [0    P    0    0 ]]

The random stream is reproduced exactly (seed protocol ``create_model.py:75-88``,
per-block ``np.random.seed(block_index)`` ``:12-21``) so that the reference's known
answer ``max_err == 0.3163456780448639`` (``examples/tests/test_examples.py:76-99``)
holds for (n_blocks=3, n_q=500, n_y_multiplier=12, n_theta=10).

Addition for benchmarking: the reference generator makes all
K_i numerically identical; ``block_values(i, k)`` perturbs the ``2I`` block of block
i at iteration k to ``(2 + eps_{i,k}) I`` so a timed run cannot factor one block and
reuse it.
"""
import numpy as np
from scipy.sparse import coo_matrix, eye

from parapint_amd.sparse.block_containers import (BlockMatrix, BlockVector,
                                                  MPIBlockMatrix, MPIBlockVector)


def distribute_blocks(num_blocks, rank, size):
    """Round-robin block -> rank map (reference utils.py:6-11)."""
    return [ndx for ndx in range(num_blocks) if ndx % size == rank]


def get_ownership_map(num_blocks, size):
    return {ndx: ndx % size for ndx in range(num_blocks)}


def random_banded(n, nnz_per_row):
    """Random n x n band matrix, values ~ N(0, 5^2) drawn in SciPy's summed-format
    data order (reference utils.py:24-31); keeps the stream bit-compatible."""
    assert nnz_per_row % 2 == 1
    m = eye(m=n, n=n, k=0, format="coo")
    for off in range(1, (nnz_per_row - 1) // 2 + 1):
        m += eye(m=n, n=n, k=off, format="coo")
        m += eye(m=n, n=n, k=-off, format="coo")
    m.data *= np.random.normal(loc=0, scale=5, size=m.data.size)
    return m


class SyntheticKKT(object):
    """Generator for N scenario blocks sharing A and P (one sparsity pattern).

    ``local_blocks`` restricts which blocks this process materialises (the MPIModel
    of the reference, create_model.py:146-255); ``None`` means all (its Model).
    """

    def __init__(self, n_blocks, n_q_per_block, n_y_multiplier, n_theta, A_nnz_per_row=3,
                 local_blocks=None):
        assert type(n_y_multiplier) is int and n_y_multiplier > 1
        self.n_blocks = n_blocks
        self.n_q = n_q_per_block
        self.n_y = n_q_per_block * n_y_multiplier
        self.n_theta = n_theta
        self.block_dim = 2 * self.n_y + self.n_q + self.n_theta
        np.random.seed(0)
        seed = np.random.randint(low=0, high=1000000)
        np.random.seed(seed)
        parts = [random_banded(n_q_per_block, A_nnz_per_row).tocoo() for _ in range(n_y_multiplier)]
        rows = np.concatenate([p.row + i * n_q_per_block for i, p in enumerate(parts)])
        cols = np.concatenate([p.col for p in parts])
        data = np.concatenate([p.data for p in parts])
        self.A = coo_matrix((data, (rows, cols)), shape=(self.n_y, self.n_q)).tocsr()
        self.theta = np.random.normal(loc=5, scale=2, size=n_theta)
        self.local_blocks = list(range(n_blocks)) if local_blocks is None else list(local_blocks)
        self.q = {}
        self.y_hat = {}
        for ndx in self.local_blocks:
            np.random.seed(ndx)
            q = np.random.normal(loc=5, scale=2, size=self.n_q)
            q[0:n_theta] = self.theta
            y_hat = self.A * q
            y_hat = y_hat + np.random.normal(loc=0, scale=0.01 * np.abs(y_hat).max(), size=self.n_y)
            self.q[ndx] = q
            self.y_hat[ndx] = y_hat
        self._build_pattern()

    # ------------------------------------------------------------------
    def _build_pattern(self):
        """COO pattern + base values of one K_i (both triangles, as the reference)."""
        n_y, n_q, n_t = self.n_y, self.n_q, self.n_theta
        o_y, o_q, o_l, o_n = 0, n_y, n_y + n_q, 2 * n_y + n_q
        A = self.A.tocoo()
        iy = np.arange(n_y)
        it = np.arange(n_t)
        rows = [o_y + iy, o_y + iy, o_q + A.col, o_q + it, o_l + iy, o_l + A.row, o_n + it]
        cols = [o_y + iy, o_l + iy, o_l + A.row, o_n + it, o_y + iy, o_q + A.col, o_q + it]
        data = [2.0 * np.ones(n_y), np.ones(n_y), -A.data, np.ones(n_t), np.ones(n_y), -A.data, np.ones(n_t)]
        self._row = np.concatenate(rows).astype(np.int32)
        self._col = np.concatenate(cols).astype(np.int32)
        self._base = np.concatenate(data).astype(np.double)
        self._n_diag = n_y   # the first n_y entries are the 2I block

    @property
    def nnz_per_block(self):
        return self._row.size

    def block_values(self, ndx, iteration=None):
        """Values of K_ndx in the fixed COO order; with ``iteration`` given, the 2I
        block becomes (2 + eps) I, eps ~ U(0, 0.5) from default_rng(10000*k + ndx)."""
        v = self._base.copy()
        if iteration is not None:
            eps = np.random.default_rng(10_000 * int(iteration) + int(ndx)).uniform(0.0, 0.5)
            v[:self._n_diag] = 2.0 + eps
        return v

    def block_matrix(self, ndx, iteration=None):
        n = self.block_dim
        return coo_matrix((self.block_values(ndx, iteration), (self._row, self._col)), shape=(n, n))

    def border_matrix(self):
        """A_i = [0, -I_{n_theta}] (reference create_model.py:104-110)."""
        n_t = self.n_theta
        it = np.arange(n_t)
        return coo_matrix((-np.ones(n_t), (it, 2 * self.n_y + self.n_q + it)), shape=(n_t, self.block_dim))

    def constant_entries(self):
        """{block: (constK, constA)}: the entries of K_i.data / A_i.data that do not depend on `iteration` -- everything but
        the (2 + eps) I block (HipSchurComplementLinearSolver.declare_constant_entries)."""
        cK = np.ones(self.nnz_per_block, dtype=bool)
        cK[:self._n_diag] = False
        cA = np.ones(self.n_theta, dtype=bool)
        return {ndx: (cK, cA) for ndx in self.local_blocks}

    def flat_values(self, iteration=None):
        """[owned blocks][nnz(K_i) + nnz(A_i)]: the values of build_kkt(iteration) as the rows of one array (K data, then the
        border's) -- parapint_amd.sparse.host_value_matrix.HostValueMatrix."""
        out = np.empty((len(self.local_blocks), self.nnz_per_block + self.n_theta))
        for i, ndx in enumerate(self.local_blocks):
            out[i, :self.nnz_per_block] = self.block_values(ndx, iteration)
        out[:, self.nnz_per_block:] = -1.0
        return out

    def block_rhs(self, ndx):
        rhs = np.zeros(self.block_dim)
        rhs[:self.n_y] = 2.0 * self.y_hat[ndx]
        return rhs

    # ------------------------------------------------------------------
    def build_kkt(self, comm=None, iteration=None, with_upper_border=True):
        """(N+1) x (N+1) block KKT.  With ``comm`` an MPIBlockMatrix carrying the
        reference's ownership table (create_model.py:207-235)."""
        N = self.n_blocks
        # (the structure of A_i does not change between iterations: one object, as an interface keeps its Jacobian)
        border = getattr(self, '_border', None)
        if border is None:
            border = self._border = self.border_matrix()
            self._border_t = border.transpose().tocoo()
        if comm is None:
            kkt = BlockMatrix(N + 1, N + 1)
        else:
            owner = -np.ones((N + 1, N + 1), dtype=np.int64)
            omap = get_ownership_map(N, comm.size)
            for ndx in range(N):
                owner[ndx, ndx] = owner[N, ndx] = owner[ndx, N] = omap[ndx]
            kkt = MPIBlockMatrix(N + 1, N + 1, owner, comm)
        for ndx in range(N):
            kkt.set_row_size(ndx, self.block_dim)
            kkt.set_col_size(ndx, self.block_dim)
        for ndx in self.local_blocks:
            kkt.set_block(ndx, ndx, self.block_matrix(ndx, iteration))
            kkt.set_block(N, ndx, border)
            if with_upper_border:
                kkt.set_block(ndx, N, self._border_t)
        kkt.set_block(N, N, coo_matrix((self.n_theta, self.n_theta)))
        return kkt

    def build_rhs(self, comm=None):
        N = self.n_blocks
        if comm is None:
            rhs = BlockVector(N + 1)
        else:
            omap = get_ownership_map(N, comm.size)
            owner = -np.ones(N + 1, dtype=np.int64)
            for ndx in range(N):
                owner[ndx] = omap[ndx]
            rhs = MPIBlockVector(N + 1, owner, comm)
        for ndx in self.local_blocks:
            rhs.set_block(ndx, self.block_rhs(ndx))
        rhs.set_block(N, np.zeros(self.n_theta))
        return rhs

    # ------------------------------------------------------------------ f2: values as device-resident sources
    def value_map(self):
        """(nsrc, src, coef) for the COO entries of K_i followed by those of A_i: what the interior-point interface of
        this problem holds per scenario are the n_y Hessian diagonal values and the nnz(A) Jacobian values
        (interfaces/interface.py:432-494: hess_block, jac_eq); every KKT entry is one of them times +-1, or the
        constant +-1 of an identity block."""
        n_y, n_t = self.n_y, self.n_theta
        nnzA = self.A.nnz
        jac = n_y + np.arange(nnzA)
        def const(k):
            return -np.ones(k)
        src = np.concatenate([np.arange(n_y), const(n_y), jac, const(n_t), const(n_y), jac, const(n_t), const(n_t)])
        coef = np.concatenate([np.ones(n_y), np.ones(n_y), -np.ones(nnzA), np.ones(n_t), np.ones(n_y), -np.ones(nnzA),
                               np.ones(n_t), -np.ones(n_t)])
        return n_y + nnzA, src.astype(np.int32), coef

    def block_sources(self, ndx, iteration=None, per_entry=False):
        """Source vector of block ndx: Hessian diagonal (2 + eps, optionally varying per entry) and Jacobian values."""
        diag = np.full(self.n_y, 2.0)
        if iteration is not None:
            eps = np.random.default_rng(10_000 * int(iteration) + int(ndx)).uniform(0.0, 0.5)
            diag = diag + eps * (np.linspace(0.5, 1.5, self.n_y) if per_entry else 1.0)
        return np.concatenate([diag, self.A.tocoo().data])

    def block_values_from_sources(self, sources):
        """Raw COO values of K_i (the order of block_values) and of A_i from a source vector, on the host."""
        nsrc, src, coef = self.value_map()
        vals = coef * np.where(src >= 0, sources[np.maximum(src, 0)], 1.0)
        return vals[:self.nnz_per_block], vals[self.nnz_per_block:]

    def build_kkt_from_sources(self, sources, comm=None):
        """Host block matrix whose K_i carry the values of the given source vectors ({block: sources})."""
        kkt = self.build_kkt(comm=comm, iteration=None)
        n = self.block_dim
        for ndx in self.local_blocks:
            kv, _ = self.block_values_from_sources(sources[ndx])
            kkt.set_block(ndx, ndx, coo_matrix((kv, (self._row, self._col)), shape=(n, n)))
        return kkt

    def build_device_kkt(self, comm=None):
        """DeviceBlockMatrix of the KKT system: host pattern (iteration 0 values) + value maps; the solver's symbolic
        phase attaches the source tensors, ``set_sources_from_host`` / the caller's kernels fill them."""
        from parapint_amd.sparse.device_containers import DeviceBlockMatrix
        nsrc, src, coef = self.value_map()
        maps = {ndx: (src, coef) for ndx in self.local_blocks}
        return DeviceBlockMatrix(self.build_kkt(comm=comm, iteration=0), maps, nsrc)

    def check_result(self, sol, comm=None):
        """max |q_est - q_true| over local blocks and |x_c - theta| (create_model.py:60-64, 134-143)."""
        max_err = 0.0
        for ndx in self.local_blocks:
            x = sol.get_block(ndx)
            x = x.flatten() if hasattr(x, 'get_block') else np.asarray(x)
            max_err = max(max_err, float(np.abs(x[self.n_y:self.n_y + self.n_q] - self.q[ndx]).max()))
        xc = sol.get_block(self.n_blocks)
        max_err = max(max_err, float(np.abs(np.asarray(xc) - self.theta).max()))
        if comm is not None and comm.size > 1:
            max_err = float(comm.allreduce_max(np.array([max_err], dtype=np.double))[0])
        return max_err
