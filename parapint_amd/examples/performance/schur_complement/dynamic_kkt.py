"""Synthetic time-staged (dynamic) KKT system with the block structure of parapint's dynamic Schur-complement
interface (BASELINE.json configs[3]).

The reference builds this structure from Pyomo.DAE models (``examples/burgers.py:63-176``, ``dynamics.py``) through
``DynamicSchurComplementInteriorPointInterface`` (interfaces/schur_complement/sc_ip_interface.py:13-1026); Pyomo is not
available, so the same *layout* (``_setup_kkt_and_rhs_structure``, :274-357) is generated for a linear-quadratic optimal
control problem cut into T time blocks:

    block t:  states x_{t,0..nfe} (n_s each), controls u_{t,0..nfe-1} (n_u each),
              dynamics x_{t,k+1} = A x_{t,k} + B u_{t,k} (equality constraints),
              cost 1/2 sum x'Q_t x + u'R_t u
    K_t  = [[ kkt_t (Hessian, Jacobian of the dynamics, explicit zero constraint diagonal),  L_bwd_t^T ],
            [ L_bwd_t,                                                                        0 * I     ]]
           L_bwd_t selects the start states x_{t,0} (their link to the coupling states z_{t-1}; absent for t = 0)
    coupling block (last block row):  rho_0..rho_{T-2} (duals of the forward links), then z_0..z_{T-2} (coupling
           states), n_c = 2 n_s (T - 1);   border (T, t) = [[ L_fwd_t at rows rho_t ], [ -I at rows z_{t-1} on the
           link duals of block t ]];   corner Q = [[0, -I], [-I, 0]]   (sc_ip_interface.py:308-357)

Every time block touches only the 2 n_s coupling rows of its own two links, so S is block-banded -- the reason the
reference builds a sparse S pattern (mpi_explicit_schur_complement.py:88-125, 228-255).
"""
import numpy as np
import scipy.sparse as sp
from scipy.sparse import coo_matrix

from parapint_amd.sparse.block_containers import BlockMatrix, BlockVector, MPIBlockMatrix, MPIBlockVector


class SyntheticDynamicKKT(object):
    def __init__(self, n_time_blocks, n_states, n_controls=2, nfe=4, seed=0, local_blocks=None):
        assert n_time_blocks >= 2
        self.T, self.n_s, self.n_u, self.nfe = n_time_blocks, n_states, n_controls, nfe
        rng = np.random.default_rng(seed)
        n_s, n_u = n_states, n_controls
        # a stable banded transition matrix (a discretised diffusion-convection operator) and a few actuators
        main = 0.6 + 0.1 * rng.random(n_s)
        A = sp.diags([0.15 + 0.05 * rng.random(n_s - 1), main, 0.1 + 0.05 * rng.random(n_s - 1)], [-1, 0, 1]).tocoo() \
            if n_s > 1 else coo_matrix(main.reshape(1, 1))
        B = np.zeros((n_s, n_u))
        for j in range(n_u):
            B[rng.integers(0, n_s), j] = 1.0
            B[:, j] += 0.1 * rng.random(n_s)
        self.A, self.B = A.tocoo(), coo_matrix(B)
        self.n_x = n_s * (nfe + 1) + n_u * nfe              # primal variables of a time block
        self.n_eq = n_s * nfe
        self.n_coupling = 2 * n_s * (self.T - 1)
        self.local_blocks = list(range(self.T)) if local_blocks is None else list(local_blocks)
        self._pattern()

    # index helpers inside a block: states of node k, controls of node k
    def _xs(self, k):
        return k * self.n_s + np.arange(self.n_s)

    def _us(self, k):
        return (self.nfe + 1) * self.n_s + k * self.n_u + np.arange(self.n_u)

    def block_dim(self, t):
        return self.n_x + self.n_eq + (self.n_s if t > 0 else 0)

    def _pattern(self):
        n_s, n_u, nfe = self.n_s, self.n_u, self.nfe
        rows, cols, vals = [], [], []
        for k in range(nfe):                                # x_{k+1} - A x_k - B u_k = 0
            r0 = k * n_s
            rows += [r0 + np.arange(n_s), r0 + self.A.row, r0 + self.B.row]
            cols += [self._xs(k + 1), self._xs(k)[self.A.col], self._us(k)[self.B.col]]
            vals += [np.ones(n_s), -self.A.data, -self.B.data]
        self.J = coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                            shape=(self.n_eq, self.n_x))

    def hessian_diagonal(self, t, iteration=None):
        """Positive weights, different for every block (and, with `iteration`, for every iteration: stands for the
        barrier terms of an interior-point iteration)."""
        rng = np.random.default_rng(7919 * (t + 1) + (0 if iteration is None else 104729 * (int(iteration) + 1)))
        return np.concatenate([1.0 + rng.random(self.n_s * (self.nfe + 1)), 0.1 + 0.05 * rng.random(self.n_u * self.nfe)])

    def block_matrix(self, t, iteration=None):
        """K_t as a nested BlockMatrix, both triangles, explicit zero diagonals where the reference has them."""
        n_x, n_eq, n_s = self.n_x, self.n_eq, self.n_s
        h = self.hessian_diagonal(t, iteration)
        kkt = BlockMatrix(2, 2)
        kkt.set_block(0, 0, coo_matrix((h, (np.arange(n_x), np.arange(n_x))), shape=(n_x, n_x)))
        kkt.set_block(1, 0, self.J)
        kkt.set_block(0, 1, self.J.transpose().tocoo())
        kkt.set_block(1, 1, coo_matrix((np.zeros(n_eq), (np.arange(n_eq), np.arange(n_eq))), shape=(n_eq, n_eq)))
        sub = BlockMatrix(2, 2)
        sub.set_block(0, 0, kkt)
        nb = n_s if t > 0 else 0
        link = BlockMatrix(1, 2)
        link.set_row_size(0, nb)
        link.set_col_size(0, n_x)
        link.set_col_size(1, n_eq)
        if t > 0:
            link.set_block(0, 0, coo_matrix((np.ones(n_s), (np.arange(n_s), self._xs(0))), shape=(n_s, n_x)))
        sub.set_block(1, 0, link)
        sub.set_block(0, 1, link.transpose())
        sub.set_block(1, 1, coo_matrix((np.zeros(nb), (np.arange(nb), np.arange(nb))), shape=(nb, nb)))
        return sub

    def border_matrix(self, t):
        """(n_c x dim K_t): forward link of block t on the rows rho_t, -I of its backward link on the rows z_{t-1}."""
        n_s, T = self.n_s, self.T
        rows, cols, vals = [], [], []
        if t < T - 1:
            rows.append(n_s * t + np.arange(n_s))
            cols.append(self._xs(self.nfe))
            vals.append(np.ones(n_s))
        if t > 0:
            rows.append(n_s * (T - 1) + n_s * (t - 1) + np.arange(n_s))
            cols.append(self.n_x + self.n_eq + np.arange(n_s))
            vals.append(-np.ones(n_s))
        return coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                          shape=(self.n_coupling, self.block_dim(t)))

    def corner_matrix(self):
        m = self.n_s * (self.T - 1)
        i = np.arange(m)
        return coo_matrix((-np.ones(2 * m), (np.concatenate([m + i, i]), np.concatenate([i, m + i]))), shape=(2 * m, 2 * m))

    def block_rhs(self, t):
        return np.random.default_rng(31 * (t + 1)).normal(size=self.block_dim(t))

    def coupling_rhs(self):
        return np.random.default_rng(5).normal(size=self.n_coupling)

    def build_kkt(self, comm=None, iteration=None):
        T = self.T
        if comm is None:
            kkt = BlockMatrix(T + 1, T + 1)
        else:
            owner = -np.ones((T + 1, T + 1), dtype=np.int64)
            for t in range(T):
                owner[t, t] = owner[T, t] = owner[t, T] = t % comm.size
            kkt = MPIBlockMatrix(T + 1, T + 1, owner, comm)
        for t in range(T):
            kkt.set_row_size(t, self.block_dim(t))
            kkt.set_col_size(t, self.block_dim(t))
        for t in self.local_blocks:
            kkt.set_block(t, t, self.block_matrix(t, iteration))
            border = self.border_matrix(t)
            kkt.set_block(T, t, border)
            kkt.set_block(t, T, border.transpose().tocoo())
        kkt.set_block(T, T, self.corner_matrix())
        return kkt

    def build_rhs(self, comm=None):
        T = self.T
        if comm is None:
            rhs = BlockVector(T + 1)
        else:
            rhs = MPIBlockVector(T + 1, np.asarray([t % comm.size for t in range(T)] + [-1]), comm)
        for t in self.local_blocks:
            rhs.set_block(t, self.block_rhs(t))
        rhs.set_block(T, self.coupling_rhs())
        return rhs

    # ------------------------------------------------------------------ f2: device-resident values
    def value_map(self, t):
        """(nsrc, src, coef) for the COO entries of K_t followed by those of A_t: the per-iteration array of this
        problem is the Hessian diagonal (n_x values per block); the Jacobian of the linear dynamics, the link matrices
        and the explicit zeros are constants.  Found by assembling the block once with marker values."""
        key = 0 if t == 0 else (2 if t == self.T - 1 else 1)
        cache = self.__dict__.setdefault('_maps', {})
        if key not in cache:
            marker = 1e9 + np.arange(self.n_x)
            real = self.hessian_diagonal
            self.hessian_diagonal = lambda tt, it=None: marker
            try:
                kd = self.block_matrix(t).tocoo().data
            finally:
                self.hessian_diagonal = real
            bd = self.border_matrix(t).tocoo().data
            vals = np.concatenate([kd, bd])
            is_src = vals > 5e8
            src = np.where(is_src, np.rint(vals - 1e9), -1).astype(np.int32)
            coef = np.where(is_src, 1.0, vals)
            cache[key] = (self.n_x, src, coef)
        return cache[key]

    def block_sources(self, t, iteration=None):
        return self.hessian_diagonal(t, iteration)

    def build_device_kkt(self, comm=None):
        from parapint_amd.sparse.device_containers import DeviceBlockMatrix
        maps = {t: self.value_map(t)[1:] for t in self.local_blocks}
        dk = DeviceBlockMatrix(self.build_kkt(comm=comm, iteration=0), maps, self.n_x)
        dk.Q = self.corner_matrix()                 # (sparse: n_c = 2 n_s (T - 1) can be tens of thousands)
        return dk
