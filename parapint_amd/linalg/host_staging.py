"""Host value staging of HipSchurComplementLinearSolver: COO blocks / flat value vectors of the caller -> the group's
canonical raw rows in page-locked staging arrays -> the device (reference: the values every sub-solver reads out of its
block at mpi_explicit_schur_complement.py:292-299; the entry-order quirk Q7 of SURVEY.md).  Fast paths: blocks whose index
arrays are the very objects verified at an earlier call are staged by the library's host threads from their two data
addresses (HipEngine.stage_upload_verified); entries the producer declared constant are skipped
(declare_constant_entries) and verified on a rotating sample.  Split out of hip_schur_complement.py (round 6); the methods
are those of the solver class."""
import numpy as np

from parapint_amd.linalg._solver_support import _PatternChanged, _S8, _addr, _coo, _index_intact, _index_record
from parapint_amd.sparse.block_containers import BlockMatrix as _BlockMatrix, MPIBlockMatrix as _MPIBlockMatrix

_OWN_MATRICES = (_BlockMatrix, _MPIBlockMatrix)     # (exact types: their get_block is a dictionary lookup)
_F8 = np.dtype(np.float64)


class HostStagingMixin(object):
    def _border(self, matrix, ndx):
        """COO of the border block A_ndx; for mapped groups with the rows in the block's local numbering."""
        A = matrix.get_block(self.block_dim - 1, ndx)
        if A is None:
            return np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32), np.zeros(0)
        br, bc, bd = _coo(A)[:3]
        if not self._mapped:
            return br, bc, bd
        bi = self._binfo[ndx]
        key = (br.__array_interface__['data'][0], br.size)
        if bi.br_cache is not None and bi.br_cache[0] == key:
            return bi.br_cache[1], bc, bd
        loc = np.searchsorted(bi.cmap, br)
        loc[loc >= max(bi.cmap.size, 1)] = 0
        if (br.size and bi.cmap.size == 0) or (br.size and np.any(bi.cmap[loc] != br)):
            raise _PatternChanged()
        loc = loc.astype(np.int32)
        bi.br_cache = (key, loc, br)                      # (the global array is kept alive with its pointer)
        return loc, bc, bd

    @staticmethod
    def _layout_positions(g, kr, kc, br, bc):
        """Canonical position of every raw entry of a layout that is not the group's reference order (-1: upper
        triangle, dropped).  Raises _PatternChanged if an entry lies outside the planned pattern."""
        keyK, keyB = g.keys()
        low = kr >= kc
        kk = kc.astype(np.int64) * g.n + kr
        posK = np.searchsorted(keyK, kk)
        posK[posK >= keyK.size] = 0
        okK = keyK[posK] == kk if keyK.size else np.zeros(kk.size, dtype=bool)
        if np.any(low & ~okK):
            raise _PatternChanged()
        posK = np.where(low, posK, -1)
        kb = br.astype(np.int64) * g.n + bc
        posB = np.searchsorted(keyB, kb)
        posB[posB >= keyB.size] = 0
        okB = keyB[posB] == kb if keyB.size else np.zeros(kb.size, dtype=bool)
        if not np.all(okB):
            raise _PatternChanged()
        return np.concatenate([posK, posB + keyK.size]).astype(np.int64)

    @classmethod
    def _canonical_values(cls, g, raw, kr, kc, br, bc, same_raw):
        """Canonical values (duplicates summed, upper triangle dropped) of one block.  Layouts other than the
        reference order (quirk Q7; a subset of the planned pattern after a re-plan) go through a cached map."""
        if same_raw:
            return np.add.reduceat(raw[g.can_idx], g.can_ptr[:-1]) if g.can_idx.size else np.zeros(0)
        pos = None
        for (akr, akc, abr, abc, apos) in g.alt_layouts:
            if (akr.size == kr.size and abr.size == br.size and np.array_equal(akr, kr) and np.array_equal(akc, kc) and
                    np.array_equal(abr, br) and np.array_equal(abc, bc)):
                pos = apos
                break
        if pos is None:
            pos = cls._layout_positions(g, kr, kc, br, bc)
            if len(g.alt_layouts) < 4:
                g.alt_layouts.append((kr.copy(), kc.copy(), br.copy(), bc.copy(), pos))
        keep = pos >= 0
        ncan = g.rowK.size + g.rowB.size
        return np.bincount(pos[keep], weights=raw[keep], minlength=ncan)

    def _stage_values(self, matrix):
        """Values of every local block into the group's compact staging array (pinned), and on to the device."""
        last = self.block_dim - 1
        fast = getattr(self._eng, 'stage_upload', None)     # threaded compare + copy + overlapped H2D in the library
        verified = getattr(self._eng, 'stage_upload_verified', None) if fast is not None else None
        batches = {}
        slow = {}
        quick = {}
        get = matrix.get_block
        binfo = self._binfo
        started = [False]
        self._stage_calls += 1
        full = self.pattern_check_interval > 0 and self._stage_calls % self.pattern_check_interval == 0
        violations = []                 # blocks whose entries declared constant changed (declare_constant_entries)
        check_now = bool(self._constant_check)
        # check=None (default): a rotating sample -- every `stride`-th block of a group, another residue at every call -- is
        # compared on the host (2 blocks of 1024 at C3: 0.2 ms); a producer that changes a "constant" for all its blocks is
        # caught at once, a single deviating block within `stride` calls (and heals at the periodic full staging either way)
        sampled = self._constant_check is None and self._constant_entries is not None
        sample_n = self.constant_sample_blocks
        sample_slots = {}
        records, memo = self._index_records, {}
        budget = [self.pattern_check_bytes]

        def intact(a):
            r = memo.get(id(a))
            if r is None:
                rec = records.get(id(a))
                chk = full
                if not chk and budget[0] >= a.nbytes:      # (index arrays shared by the blocks: checked at every call)
                    budget[0] -= a.nbytes
                    chk = True
                r = memo[id(a)] = rec is not None and rec[0] is a and _index_intact(a, rec, chk)
            return r

        def flush(q):
            g, slots, kps, bps = q
            if not slots:
                return
            if any(slots[i] >= slots[i + 1] for i in range(len(slots) - 1)):
                order = sorted(range(len(slots)), key=slots.__getitem__)
                slots, kps, bps = [slots[i] for i in order], [kps[i] for i in order], [bps[i] for i in order]
            started[0] = True
            self._send_verified(g, slots, kps, bps, full)
            del q[1][:], q[2][:], q[3][:]

        # (blocks of the package's own containers: the dictionary behind get_block, one call frame less per block)
        table = getattr(matrix, '_blocks', None) if type(matrix) in _OWN_MATRICES else None
        tget = table.get if isinstance(table, dict) else None
        setok = {}                      # id(tuple of a block's four index arrays) -> all four are intact
        nflush = 64
        try:
            for ndx in self.local_block_indices:
                bi = binfo[ndx]
                g = bi.group
                K = tget((ndx, ndx)) if tget is not None else get(ndx, ndx)
                c = bi.seen
                if c is not None and verified is not None:
                    # the block's index arrays are the objects an earlier call compared with the group's reference order
                    # (typical: the interface rewrites .data of the same COO blocks at every iteration, or hands out new
                    # blocks over shared index arrays): only the two data addresses are needed.  The index arrays are
                    # checked for having been rewritten in place: size and address at every call, a checksum of their
                    # contents at every `pattern_check_interval`-th call (once per array object and call; blocks that
                    # share all four arrays share the answer).
                    A = tget((last, ndx)) if tget is not None else get(last, ndx)
                    try:
                        # (.coords: the (row, col) tuple of a SciPy >= 1.13 COO block; .row / .col are properties there)
                        try:
                            ck, ca = K.coords, A.coords
                        except AttributeError:
                            ck, ca = (K.row, K.col), (A.row, A.col)
                        hit = ck[0] is c[0] and ck[1] is c[1] and ca[0] is c[2] and ca[1] is c[3]
                    except AttributeError:
                        hit = False
                    if hit:
                        kd, bd = K.data, A.data
                        if kd.size == c[4] and bd.size == c[5] and kd.dtype is _F8 and bd.dtype is _F8 and \
                                kd.strides == _S8 and bd.strides == _S8:
                            cs = c[6]
                            ok = setok.get(id(cs))
                            if ok is None:
                                ok = setok[id(cs)] = intact(cs[0]) and intact(cs[1]) and intact(cs[2]) and intact(cs[3])
                            if ok:
                                look = check_now
                                if sampled and not look:
                                    ss = sample_slots.get(g.gid)
                                    if ss is None:          # (this call's residue class of the group's slots)
                                        stride = max(1, len(g.blocks) // max(sample_n, 1))
                                        ss = sample_slots[g.gid] = frozenset(range(self._stage_calls % stride, len(g.blocks), stride)) \
                                            if sample_n > 0 else frozenset()
                                    look = bi.slot in ss
                                if look and g.const_src is not None and g.full_rows is not None and g.full_rows[bi.slot]:
                                    # (check=True: the entries declared constant against the staging row -- a debugging aid)
                                    sK, dK, sB, dB = g.const_parts
                                    row = g.staging[bi.slot]
                                    if not (np.array_equal(kd[sK], row[dK]) and np.array_equal(bd[sB], row[dB])):
                                        violations.append(ndx)
                                        g.full_rows[bi.slot] = False         # (staged over every entry below)
                                q = quick.get(g.gid)
                                if q is None:
                                    q = quick[g.gid] = (g, [], [], [])
                                q[1].append(bi.slot)
                                q[2].append(_addr(kd))
                                q[3].append(_addr(bd) if c[5] else 0)
                                if len(q[1]) == nflush:
                                    # on its way while the next blocks are looked at (the library's host threads stage and send)
                                    flush(q)
                                    nflush = 256
                                continue
                    bi.seen = None
                kr, kc, kd, _ = _coo(K)
                br, bc, bd = self._border(matrix, ndx)
                arrays = (kr, kc, kd, br, bc, bd)
                if fast is not None and all(a.flags.c_contiguous for a in arrays) and \
                        kr.dtype == kc.dtype == br.dtype == bc.dtype == np.int32 and kd.dtype == bd.dtype == np.float64:
                    batches.setdefault(g.gid, (g, []))[1].append((bi.slot, arrays, bi, K))
                else:
                    self._stage_block(g, bi.slot, *arrays)
                    slow.setdefault(g.gid, (g, []))[1].append(bi.slot)
            for q in quick.values():
                flush(q)
        finally:
            if started[0]:
                self._eng.stage_upload_end()            # (also on the way out with a changed pattern: no job stays in flight)
        for g, items in batches.values():
            items.sort(key=lambda it: it[0])
            same = fast(g, [it[:2] for it in items], full)
            for ok, (slot, arrays, bi, K) in zip(same, items):
                if ok:
                    g.full_rows[slot] = True             # (the library staged every entry of the row)
                if not ok:
                    self._stage_block(g, slot, *arrays)
                    slow.setdefault(g.gid, (g, []))[1].append(slot)
                elif verified is not None and getattr(K, 'format', None) == 'coo':
                    A = get(last, g.blocks[slot])
                    if getattr(A, 'format', None) == 'coo' and K.row is arrays[0] and K.col is arrays[1] and \
                            A.col is arrays[4] and (A.row is arrays[3] or (bi.br_cache is not None and A.row is bi.br_cache[2])):
                        four = (K.row, K.col, A.row, A.col)
                        four = self._index_sets.setdefault(tuple(map(id, four)), four)      # one tuple per set of arrays
                        bi.seen = four + (arrays[2].size, arrays[5].size, four)
                        for a in bi.seen[:4]:
                            if not memo.get(id(a), False):       # (verified equal to the reference order just now)
                                records[id(a)] = _index_record(a)
                                memo[id(a)] = True
        if fast is None:
            for g in self._groups:
                self._eng.upload_values_compact(g.gid, g.staging)
        else:
            for g, slots in slow.values():                   # rows the library did not stage itself
                for slot in slots:
                    self._eng.upload_values_compact(g.gid, g.staging, slot, 1)
        self._report_constant_violations(violations)

    def _report_constant_violations(self, violations):
        if violations:
            # (every entry of these blocks was staged all the same: the factorisation that follows is of the matrix handed over)
            err = RuntimeError('staging: entries declared constant (declare_constant_entries) have changed in block(s) %s'
                               % violations[:8])
            err.status = 3
            raise err

    def _send_verified(self, g, slots, kps, bps, every_entry):
        """One batch of verified blocks to the library's staging threads: over the entries not declared constant for rows
        whose staging row holds every entry of its block already, over all entries otherwise (and, every_entry, for all rows:
        the periodic pass that lets a declaration that does not hold heal)."""
        verified = self._eng.stage_upload_verified
        _ = g.staging                                       # (allocates the rows and their flags)
        fr = g.full_rows
        idx = np.asarray(slots, dtype=np.int64)
        if g.var_runs is None or every_entry:
            verified(g, slots, kps, bps)
            fr[idx] = True
            return
        have = fr[idx]
        if have.all():
            verified(g, slots, kps, bps, g.var_runs)
        elif not have.any():
            verified(g, slots, kps, bps)
            fr[idx] = True
        else:
            kps, bps = np.asarray(kps, dtype=np.uint64), np.asarray(bps, dtype=np.uint64)
            verified(g, idx[~have], kps[~have], bps[~have])           # (a second begin waits for the first job)
            fr[idx[~have]] = True
            verified(g, idx[have], kps[have], bps[have], g.var_runs)

    def _stage_flat_values(self, matrix):
        """Values of a HostValueMatrix (one flat vector per block: K data then A data, over the pattern object the symbolic
        phase saw) into the staging arrays and on to the device.  A pattern group whose blocks all come in the group's
        reference entry order is staged by ONE library call when the vectors are the rows of one 2-D array (addresses by
        arithmetic), else with one address per block; blocks in another entry order (after a re-plan on a union pattern)
        are canonicalised on the host like any other block."""
        pat = matrix.pattern
        if pat is not self._symbolic_pattern:
            raise RuntimeError('this HostValueMatrix is not over the matrix given to do_symbolic_factorization')
        vals = matrix.flat_values
        verified = getattr(self._eng, 'stage_upload_verified', None)
        self._stage_calls += 1
        full = self.pattern_check_interval > 0 and self._stage_calls % self.pattern_check_interval == 0
        two_d = isinstance(vals, np.ndarray)
        if two_d:
            if vals.ndim != 2 or vals.dtype != _F8 or vals.strides[1] != 8 or vals.shape[0] != len(self.local_block_indices):
                raise ValueError('flat_values: a 2-D array must be float64 [owned blocks][entries] with contiguous rows')
            rows_of = self._flat_rows.get('rows')
            if rows_of is None:
                rows_of = self._flat_rows['rows'] = {ndx: i for i, ndx in enumerate(self.local_block_indices)}
        started = False
        slow = []
        violations = []
        try:
            for g in self._groups:
                nK, nB = g.nrawK, g.nraw - g.nrawK
                cached = self._flat_rows.get(g.gid)
                if cached is None or cached[0] is not g:
                    same = all(self._binfo[ndx].raw_sig for ndx in g.blocks)
                    rows = np.array([rows_of[ndx] for ndx in g.blocks], dtype=np.int64) if two_d else None
                    cached = self._flat_rows[g.gid] = (g, same, rows)
                _, same, rows = cached
                if two_d and rows is None:
                    rows = np.array([rows_of[ndx] for ndx in g.blocks], dtype=np.int64)
                    self._flat_rows[g.gid] = (g, same, rows)
                if verified is None or not same or (two_d and vals.shape[1] != g.nraw):
                    slow.append(g)
                    continue
                if two_d:
                    kd = (vals.ctypes.data + rows * vals.strides[0]).astype(np.uint64)
                else:
                    addrs = []
                    for ndx in g.blocks:
                        v = vals[ndx]
                        if type(v) is not np.ndarray or v.dtype is not _F8 or v.strides != _S8 or v.size != g.nraw:
                            addrs = None
                            break
                        addrs.append(_addr(v))
                    if addrs is None:
                        slow.append(g)
                        continue
                    kd = np.array(addrs, dtype=np.uint64)
                bd = kd + np.uint64(8 * nK) if nB else np.zeros(kd.size, dtype=np.uint64)
                started = True
                sampled = self._constant_check is None and self._constant_entries is not None and self.constant_sample_blocks > 0
                if (self._constant_check or sampled) and g.const_src is not None and g.full_rows is not None:
                    src, dst = g.const_src, g.const_dst
                    stride = max(1, len(g.blocks) // self.constant_sample_blocks) if sampled else 1
                    for slot, ndx in enumerate(g.blocks):
                        if sampled and slot % stride != self._stage_calls % stride:
                            continue
                        if g.full_rows[slot]:
                            v = vals[rows[slot]] if two_d else vals[ndx]
                            if not np.array_equal(v[src], g.staging[slot][dst]):
                                violations.append(ndx)
                                g.full_rows[slot] = False
                self._send_verified(g, np.arange(len(g.blocks), dtype=np.int32), kd, bd, full)
        finally:
            if started:
                self._eng.stage_upload_end()
        last = self.block_dim - 1
        for g in slow:
            nK = None
            for slot, ndx in enumerate(g.blocks):
                v = vals[rows_of[ndx]] if two_d else vals[ndx]
                v = np.ascontiguousarray(v, dtype=np.double).ravel()
                kr, kc, kd0, _ = _coo(pat.get_block(ndx, ndx))
                br, bc, _bd = self._border(pat, ndx)
                if v.size != kd0.size + _bd.size:
                    raise ValueError('flat_values: block %d has %d entries, its pattern has %d' % (ndx, v.size, kd0.size + _bd.size))
                self._stage_block(g, slot, kr, kc, v[:kd0.size], br, bc, v[kd0.size:])
            self._eng.upload_values_compact(g.gid, g.staging)
        self._report_constant_violations(violations)

    def _stage_block(self, g, slot, kr, kc, kd, br, bc, bd):
        ref = g.raw_refs
        same = (kd.size == g.nrawK and bd.size == g.nraw - g.nrawK and
                (kr is ref[0] or np.array_equal(kr, ref[0])) and (kc is ref[1] or np.array_equal(kc, ref[1])) and
                (br is ref[2] or np.array_equal(br, ref[2])) and (bc is ref[3] or np.array_equal(bc, ref[3])))
        row = g.staging[slot]
        g.full_rows[slot] = True                    # (every entry of the block is written below)
        if same:
            for e0, ln, dst in g.runsK:
                row[dst:dst + ln] = kd[e0:e0 + ln]
            for e0, ln, dst in g.runsB:
                row[dst:dst + ln] = bd[e0:e0 + ln]
        else:
            vals = self._canonical_values(g, np.concatenate([kd, bd]), kr, kc, br, bc, False)
            row[:] = 0.0
            row[g.can_cidx[g.can_ptr[:-1]]] = vals     # canonical sum on the first raw slot of each entry

    def declare_constant_entries(self, constant, check=None):
        """An interface whose Jacobian or Hessian values do not change between numeric factorisations (linear constraints, a
        QP: the reference's interfaces hand all of them over again at every iteration, parapint/interfaces/interface.py:
        evaluate_primal_dual_kkt_matrix) says so here, after do_symbolic_factorization; the declaration holds (over re-plans
        and pivot-order refreshes, too) until it is withdrawn (constant = None) or the next do_symbolic_factorization.

        constant: {block index: (constK, constA)} -- bool per entry of the block's K_ii.data and of its border A_i.data, in
        the order of the blocks handed to do_symbolic_factorization (None / missing block: nothing constant).  A pattern
        group takes the entries constant in ALL its blocks; groups whose blocks come in different entry orders ignore the
        declaration.  Effect: for a block whose staging row already holds all its entries the library's staging threads are
        given the runs of the OTHER entries only (pp_stage_upload_verified_begin: compare and copy) -- host COO blocks or flat
        value vectors in; the device interface has its value maps for that.  At every `pattern_check_interval`-th call every
        entry is staged again (a declaration that does not hold heals there).  check=None (default): at every call the declared
        entries of a rotating sample of blocks (`constant_sample_blocks` = 2 per group) are compared with the staging rows on the
        host -- a producer that changes a "constant" for all its blocks is caught at the first call; check=True: all blocks at
        every call (a debugging aid); check=False: none.  What changed is staged in full and the factorisation returns an error
        status that names the blocks."""
        if not getattr(self, '_groups', None) or getattr(self, 'plan_stats', None) is None:
            raise RuntimeError('declare_constant_entries: call do_symbolic_factorization first')
        self._constant_entries = None if constant is None else dict(constant)
        self._constant_check = check
        self._apply_constant_entries()

    def _apply_constant_entries(self):
        decl = self._constant_entries
        for g in self._groups:
            runs = None
            if decl is not None and all(self._binfo[ndx].raw_sig and decl.get(ndx) is not None for ndx in g.blocks):
                nK, nB = g.nrawK, g.nraw - g.nrawK
                mask = np.ones(g.nraw, dtype=bool)
                for ndx in g.blocks:
                    cK, cA = decl[ndx]
                    cK = np.zeros(nK, dtype=bool) if cK is None else np.asarray(cK, dtype=bool).ravel()
                    cA = np.zeros(nB, dtype=bool) if cA is None else np.asarray(cA, dtype=bool).ravel()
                    if cK.size != nK or cA.size != nB:
                        mask = None
                        break
                    mask[:nK] &= cK
                    mask[nK:] &= cA
                if mask is not None:
                    runs = g.variable_runs(mask)
            if runs is None:
                g.const_src = g.const_dst = None
            g.var_runs = runs

    def _stage_and_upload(self, matrix):
        changed = 0
        failed = None
        flat = hasattr(matrix, 'flat_values') and matrix.flat_values is not None
        try:
            if flat:
                self._stage_flat_values(matrix)
            else:
                self._stage_values(matrix.pattern if hasattr(matrix, 'flat_values') else matrix)
        except _PatternChanged:
            changed = 1
        except Exception as err:
            # a failure that carries a C status (a violated constant-entry declaration, a staging call that failed) on
            # THIS rank: the others are on their way into the collective below -- join it first, fail afterwards (the
            # status then reaches the all-reduce of S through _guarded / fail_local like any other host-side failure)
            if getattr(err, 'status', None) is None or self.comm.size == 1:
                raise
            failed = err
        if self.comm.size > 1:
            # the new plan is made collectively (its coupling structure is agreed by all ranks): a rank whose own
            # blocks still fit the old pattern re-plans with the others
            changed = int(self.comm.allreduce_max(np.array([changed], dtype=np.int64))[0])
        if failed is not None and not changed:
            raise failed
        if changed:
            # entries outside the planned pattern (the inertia-correction loop adds diagonal blocks): plan again
            # on the union of both patterns, as the reference's MUMPS sub-solver does (mumps_interface.py:82-83)
            if flat:
                raise RuntimeError('HostValueMatrix: the pattern object was modified after the symbolic factorisation')
            self._replan_union(matrix)
            self._stage_values(matrix)
        if self._pattern_only:
            # symbolic saw no usable values (quirk Q8): fix the pivot sequence now
            for g in self._groups:
                if g.rep_vals is None:
                    g.rep_vals = g.canonical_from_compact(g.staging[0])
            self._run_symbolic()
            self._pattern_only = False
            for g in self._groups:                           # (the new plan's device buffers are empty)
                self._eng.upload_values_compact(g.gid, g.staging)
