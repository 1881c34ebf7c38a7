"""Structure of the coupling block S for mapped pattern groups (mixin of HipSchurComplementLinearSolver): dense, or an
ordering and a block size under which S is block tridiagonal (the time blocks of a dynamic problem), and the
conversions between the caller's coupling order and the library's (reference: the sparse S pattern of
mpi_explicit_schur_complement.py:88-125, 228-255)."""
import numpy as np


class CouplingStructureMixin(object):
    def _coupling_structure(self, matrix, groups):
        """Dense S, or -- for mapped groups whose cliques form a band (the time blocks of a dynamic problem only touch
        the coupling variables of their own two links) -- an ordering and a block size under which S is block
        tridiagonal.  The reference keeps S sparse for the same reason (mpi_...:88-125, 228-255).  Collective."""
        nc = self._nc
        self._cperm = self._cinv = None
        self._btd = None
        if not self._mapped or not getattr(self._eng, 'supports_block_tridiagonal', False) or nc <= self._dense_coupling_limit:
            return
        from scipy.sparse import coo_matrix
        from scipy.sparse.csgraph import reverse_cuthill_mckee
        # every rank needs the cliques of all blocks: one sum all-reduce of a [blocks][m_max + 1] table
        nb = self.block_dim - 1
        mmax = max([g.m for g in groups] + [0])
        if self.comm.size > 1:
            mmax = int(self.comm.allreduce_max(np.array([mmax], dtype=np.int64))[0])
        table = np.zeros((nb, mmax + 1), dtype=np.int64)
        for g in groups:
            for ndx, cm in zip(g.blocks, g.cmaps):
                table[ndx, 0] = cm.size
                table[ndx, 1:1 + cm.size] = cm + 1
        if self.comm.size > 1:
            table = self.comm.allreduce_sum(table.astype(np.double)).astype(np.int64)
        cliques = [table[ndx, 1:1 + table[ndx, 0]] - 1 for ndx in range(nb)]
        Qb = matrix.get_block(self.block_dim - 1, self.block_dim - 1)
        Qc = Qb.tocoo() if Qb is not None else None
        self._btd_sequential = False
        # (1) natural blocks: the coupling rows every time block touches (merged where blocks overlap) are the diagonal
        # blocks of S; if Q only links consecutive ones, S is block tridiagonal in that order.  These blocks are what the
        # problem's own structure makes well-posed (a block's clique is the Schur contribution of ONE K_t), so that the
        # odd-even elimination order of cyclic reduction meets no singular diagonal block.
        parent = np.arange(nc)

        def find(a):
            while parent[a] != a:
                parent[a] = parent[parent[a]]
                a = parent[a]
            return a
        for cm in cliques:
            for v in cm[1:]:
                ra, rb = find(int(cm[0])), find(int(v))
                if ra != rb:
                    parent[rb] = ra
        root = np.array([find(i) for i in range(nc)])
        atoms, atom_of = np.unique(root, return_inverse=True)
        na = atoms.size
        if 3 <= na and Qc is not None:
            a_r, a_c = atom_of[Qc.row], atom_of[Qc.col]
            off = a_r != a_c
            AG = coo_matrix((np.ones(int(off.sum()) + na), (np.concatenate([a_r[off], np.arange(na)]),
                                                           np.concatenate([a_c[off], np.arange(na)]))), shape=(na, na)).tocsr()
            aperm = np.asarray(reverse_cuthill_mckee(AG, symmetric_mode=True), dtype=np.int64)
            apos = np.empty(na, dtype=np.int64)
            apos[aperm] = np.arange(na)
            path = (not off.any()) or int(np.abs(apos[a_r[off]] - apos[a_c[off]]).max()) <= 1
            if path:
                # blocks straddle the cliques: a variable linked by Q to the NEXT clique opens a block, one linked to the
                # PREVIOUS clique closes the block before -- block p = (forward-linked part of clique p) + (backward-
                # linked part of clique p + 1), i.e. the pairs Q ties together (for a time-staged problem: the duals of the
                # forward links of block t with the coupling states z_t).  Diagonal blocks that contain such pairs stay
                # well conditioned under any elimination order; the cliques themselves do not (a clique block is a
                # principal submatrix of inv(K_t), rank deficient up to rounding when a time block has few controls).
                pos_v = apos[atom_of]
                back = np.zeros(nc, dtype=bool)
                d = apos[a_c] - apos[a_r]
                back[Qc.row[d == -1]] = True                       # a Q partner in the previous clique
                fwd = np.zeros(nc, dtype=bool)
                fwd[Qc.row[d == 1]] = True
                blk = np.where(back & ~fwd, pos_v - 1, pos_v)
                used, blk = np.unique(blk, return_inverse=True)    # drop empty blocks, keep the order
                G = used.size
                ok = True
                for cm in cliques:
                    if cm.size and int(blk[cm].max() - blk[cm].min()) > 1:
                        ok = False
                if ok and int(np.abs(blk[Qc.row] - blk[Qc.col]).max()) <= 1 and G >= 3:
                    sizes = np.bincount(blk, minlength=G)
                    gs = int(sizes.max())
                    if gs <= 512:
                        order = np.argsort(blk, kind='stable')
                        start = np.concatenate([[0], np.cumsum(sizes)])
                        within = np.zeros(nc, dtype=np.int64)
                        within[order] = np.arange(nc) - start[blk[order]]
                        inv = blk * gs + within
                        pad_map = -np.ones(G * gs, dtype=np.int64)
                        pad_map[inv] = np.arange(nc)
                        self._cperm, self._cinv, self._cperm_pad = pad_map[pad_map >= 0], inv, pad_map
                        self._btd = (gs, G)
                        return
        # (2) otherwise: a bandwidth-reducing ordering cut into blocks of the bandwidth, eliminated in ascending order
        rows, cols = [np.arange(nc)], [np.arange(nc)]
        for cm in cliques:
            rows.append(np.repeat(cm, cm.size))
            cols.append(np.tile(cm, cm.size))
        if Qc is not None:
            rows += [Qc.row, Qc.col]
            cols += [Qc.col, Qc.row]
        rows, cols = np.concatenate(rows), np.concatenate(cols)
        P = coo_matrix((np.ones(rows.size), (rows, cols)), shape=(nc, nc)).tocsr()
        perm = np.asarray(reverse_cuthill_mckee(P, symmetric_mode=True), dtype=np.int64)     # new -> old
        inv = np.empty(nc, dtype=np.int64)
        inv[perm] = np.arange(nc)
        hb = int(np.abs(inv[rows] - inv[cols]).max())
        gs = max(hb, 1)
        G = -(-nc // gs)
        if gs > 512 or G < 3:
            return                                      # not banded enough: dense S
        self._cperm, self._cinv = perm, inv
        self._cperm_pad = np.concatenate([perm, -np.ones(gs * G - nc, dtype=np.int64)])     # new (padded) -> old, -1: padding
        self._btd = (gs, G)
        self._btd_sequential = True

    def _btd_corner(self, Q):
        """Symmetric sparse Q (or None) -> (positions, values) in the block-tridiagonal layout of the Schur buffer in the
        permuted order (duplicates add), with a unit diagonal on the padding rows."""
        gs, G = self._btd
        g2 = gs * gs
        pad = np.flatnonzero(self._cperm_pad < 0)
        pos = [(pad // gs) * g2 + (pad % gs) * (gs + 1)]
        val = [np.ones(pad.size)]
        if Q is not None:
            from scipy.sparse import coo_matrix as _coo_m
            Qc = _coo_m(Q)
            i, j, v = Qc.row, Qc.col, Qc.data
            pi, pj = self._cinv[i], self._cinv[j]
            bi_, bj_ = pi // gs, pj // gs
            same = bi_ == bj_
            pos.append(bi_[same] * g2 + (pi[same] % gs) + (pj[same] % gs) * gs)
            val.append(v[same])
            low = bi_ == bj_ + 1                          # E_t = S(block t+1, block t): only this orientation is stored
            pos.append(G * g2 + bj_[low] * g2 + (pi[low] % gs) + (pj[low] % gs) * gs)
            val.append(v[low])
            if np.any(~same & ~low & (bj_ != bi_ + 1)):
                raise RuntimeError('coupling block Q has entries outside the block-tridiagonal structure')
        return np.concatenate(pos).astype(np.int64), np.concatenate(val).astype(np.float64)

    def _btd_q(self, Q):
        """The same as one flat array in the layout of the Schur buffer (tests, host interpreter)."""
        gs, G = self._btd
        flat = np.zeros((2 * G - 1) * gs * gs)
        pos, val = self._btd_corner(Q)
        np.add.at(flat, pos, val)
        return flat

    def _to_coupling_order(self, v):
        """Coupling vector in the caller's order -> the library's (permuted, padded) order."""
        if self._btd is None:
            return v
        out = np.zeros(self._btd[0] * self._btd[1])
        out[self._cinv] = v
        return out
