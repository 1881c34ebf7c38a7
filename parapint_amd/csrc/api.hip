// Handle life cycle, symbolic phase (plan -> device images), value / vector uploads and bindings, statistics, host staging.
#include "common.hpp"
#include <dlfcn.h>

// ncclUniqueId by value (rccl.h: struct { char internal[128]; })
struct ppd_nccl_id { char internal[128]; };
#include "kernels_transpose.hpp"

namespace {


}  // namespace

namespace { void rccl_release(pp_handle h); }

extern "C" {

static std::string g_create_error;

int pp_create(pp_handle* out, int device, void* stream) {
  if (!out) return 3;
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_create_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e) + " (devices: " + std::to_string(ndev) + ")";
    return 3;
  }
  pp_handle h = new pp_solver();
  if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
  h->device = device;
  e = hipSetDevice(device);
  if (e != hipSuccess) {
    g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
    delete h;
    return 3;
  }
  h->stream = (hipStream_t)stream;
  *out = h;
  return 0;
}

namespace { int stage_job_finish(pp_handle h); }

void pp_destroy(pp_handle h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)stage_job_finish(h);
  (void)join_dense(h);
  (void)hipStreamSynchronize(h->stream);
  for (Group* g : h->groups) free_group(g);
  free_globals(h);
  if (h->ev_made)
    for (int i = 0; i < PP_NPHASE; ++i) { (void)hipEventDestroy(h->ev[i][0]); (void)hipEventDestroy(h->ev[i][1]); }
  for (hipEvent_t e : h->dl_events) (void)hipEventDestroy(e);
  if (h->ev_corner_up) { (void)hipEventDestroy(h->ev_corner_up); (void)hipEventDestroy(h->ev_corner_done); (void)hipStreamDestroy(h->up_stream); }
  if (h->aux_made) {
    for (int i = 0; i < PP_MAX_SPLIT; ++i) { (void)hipStreamDestroy(h->aux[i]); (void)hipEventDestroy(h->ev_join[i]); }
    (void)hipEventDestroy(h->ev_fork);
  }
  if (h->ev_blocks_done) (void)hipEventDestroy(h->ev_blocks_done);
  if (h->dense_stream) {
    (void)hipStreamSynchronize(h->dense_stream);
    (void)hipStreamDestroy(h->dense_stream);
    (void)hipEventDestroy(h->ev_dense_fork);
    (void)hipEventDestroy(h->ev_dense_done);
  }
  if (h->ev_coll_side) (void)hipEventDestroy(h->ev_coll_side);
  if (h->ip_part) (void)hipFree(h->ip_part);
  if (h->ip_cmax) (void)hipFree(h->ip_cmax);
  if (h->ip_mail_host) (void)hipHostFree((void*)h->ip_mail_host);
  rccl_release(h);
  delete h;
}

const char* pp_last_error(pp_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int pp_begin_symbolic(pp_handle h, int n_coupling) {
  if (!h) return 3;
  if (n_coupling < 0) return fail(h, 3, "negative coupling dimension");
  PP_HIP(hipSetDevice(h->device));
  if (int rc = join_dense(h)) return rc;
  PP_HIP(hipStreamSynchronize(h->stream));
  for (Group* g : h->groups) free_group(g);
  h->groups.clear();
  free_globals(h);
  h->nc = n_coupling;
  h->btd = 0; h->gs = h->G = 0;
  h->symbolic_done = h->blocks_factored = h->numeric_done = h->schur_done = false;
  return 0;
}

int pp_set_coupling_structure(pp_handle h, int mode, int gs, int G) {
  if (!h) return 3;
  if (h->symbolic_done || !h->groups.empty()) return fail(h, 3, "pp_set_coupling_structure: call right after pp_begin_symbolic");
  if (mode == 0) { h->btd = 0; h->gs = h->G = 0; return 0; }
  if (mode != 1 || gs < 1 || gs > 512 || G < 1 || (int64_t)gs * G != h->nc)
    return fail(h, 3, "pp_set_coupling_structure: mode 1 needs n_c = G * gs, 1 <= gs <= 512");
  h->btd = 1; h->gs = gs; h->G = G;
  return 0;
}

int pp_set_coupling_schedule(pp_handle h, int sequential) {
  if (!h) return 3;
  h->btd_sequential = sequential ? 1 : 0;
  if (h->symbolic_done && h->btd) {
    PP_HIP(hipSetDevice(h->device));
    PP_HIP(hipStreamSynchronize(h->stream));
    return build_btd_schedule(h);
  }
  return 0;
}

int64_t pp_schur_buffer_doubles(pp_handle h) { return h ? (int64_t)(schur_doubles(h) + PP_TAIL) : 0; }

int pp_add_group(pp_handle h, int n, int batch, int nnzK, const int32_t* rowK, const int32_t* colK, int nnzB,
                 const int32_t* rowB, const int32_t* colB, int nraw, const int32_t* can_ptr, const int32_t* can_idx,
                 const double* rep_vals, int* group_out) {
  if (!h) return 3;
  return pp_add_group_mapped(h, n, batch, nnzK, rowK, colK, nnzB, rowB, colB, nraw, can_ptr, can_idx, rep_vals, h->nc,
                             nullptr, group_out);
}

int pp_add_group_mapped(pp_handle h, int n, int batch, int nnzK, const int32_t* rowK, const int32_t* colK, int nnzB,
                        const int32_t* rowB, const int32_t* colB, int nraw, const int32_t* can_ptr, const int32_t* can_idx,
                        const double* rep_vals, int nc_loc, const int32_t* cmap, int* group_out) {
  if (!h) return 3;
  if (nc_loc < 0 || nc_loc > h->nc) return fail(h, 3, "pp_add_group_mapped: local coupling dimension out of range");
  if (!cmap && nc_loc != h->nc) return fail(h, 3, "pp_add_group_mapped: a group without a map uses all coupling rows");
  if (cmap)
    for (size_t i = 0; i < (size_t)batch * nc_loc; ++i)
      if (cmap[i] < 0 || cmap[i] >= h->nc) return fail(h, 3, "pp_add_group_mapped: coupling map entry out of range");
  if (h->symbolic_done) return fail(h, 3, "pp_add_group after pp_end_symbolic");
  if (batch <= 0 || n <= 0 || nraw < 0) return fail(h, 3, "bad group dimensions");
  Group* g = new Group();
  pp::PlanOptions opt;
  pp::tune_for_batch(opt, batch);
  if (cmap) pp::tune_for_mapped_group(opt, batch);
  if (h->sn_wmax > 0) opt.sn_wmax = h->sn_wmax;
  if (h->sn_tol >= 0) opt.sn_tol_rows = h->sn_tol;
  if (h->pivot_threshold > 0.0) opt.pivot_threshold = h->pivot_threshold;
  {
    std::string bad;
    if (!pp::apply_plan_tune(opt, pp::env_switch("PP_PLAN_TUNE"), bad)) { delete g; return fail(h, 3, "PP_PLAN_TUNE: unknown key " + bad); }
  }
  g->nc_loc = nc_loc;
  if (cmap) g->cmap_host.assign(cmap, cmap + (size_t)batch * nc_loc);
  const int ncan = nnzK + nnzB;
  g->diag_can.assign((size_t)n, -1);
  for (int e = 0; e < nnzK; ++e)
    if (rowK[e] == colK[e]) g->diag_can[(size_t)rowK[e]] = e;
  g->batch = batch; g->nraw = nraw;
  g->can_ptr.assign(can_ptr, can_ptr + ncan + 1);
  g->can_idx.assign(can_idx, can_idx + can_ptr[ncan]);
  for (int v : g->can_idx)
    if (v < 0 || v >= nraw) { delete g; return fail(h, 3, "canonical map points outside the raw vector"); }
  // the plan itself (ordering, pivot sequence, schedule: the expensive part) is built by pp_end_symbolic, all groups of the
  // handle side by side on host threads -- three pattern groups of a time-staged problem take the time of one
  PendingPlan* pn = new PendingPlan();
  pn->n = n; pn->opt = opt;
  pn->rowK.assign(rowK, rowK + nnzK); pn->colK.assign(colK, colK + nnzK);
  pn->rowB.assign(rowB, rowB + nnzB); pn->colB.assign(colB, colB + nnzB);
  if (rep_vals) pn->rep_vals.assign(rep_vals, rep_vals + ncan);
  pn->have_vals = rep_vals != nullptr;
  g->pending = pn;
  h->groups.push_back(g);
  if (group_out) *group_out = (int)h->groups.size() - 1;
  return 0;
}

int pp_end_symbolic(pp_handle h) {
  if (!h) return 3;
  PP_HIP(hipSetDevice(h->device));
  const int nc = h->nc;
  {
    std::vector<Group*> todo;
    for (Group* g : h->groups) if (g->pending) todo.push_back(g);
    std::vector<int> rcs(todo.size(), 0);
    auto build = [&](size_t i) {
      Group* g = todo[i];
      PendingPlan& pn = *g->pending;
      rcs[i] = pp::build_plan(pn.n, g->nc_loc, (int)pn.rowK.size(), pn.rowK.data(), pn.colK.data(), (int)pn.rowB.size(),
                              pn.rowB.data(), pn.colB.data(), pn.have_vals ? pn.rep_vals.data() : nullptr, pn.opt, g->plan);
    };
    if (todo.size() > 1) {
      std::vector<std::thread> th;
      for (size_t i = 1; i < todo.size(); ++i) th.emplace_back(build, i);
      build(0);
      for (auto& t : th) t.join();
    } else if (todo.size() == 1) build(0);
    for (size_t i = 0; i < todo.size(); ++i) {
      // (the canonical pattern stays with the group: the residual rows of the a-posteriori check are built from it)
      todo[i]->pat_rowK.swap(todo[i]->pending->rowK); todo[i]->pat_colK.swap(todo[i]->pending->colK);
      todo[i]->pat_rowB.swap(todo[i]->pending->rowB); todo[i]->pat_colB.swap(todo[i]->pending->colB);
      delete todo[i]->pending; todo[i]->pending = nullptr;
    }
    for (size_t i = 0; i < todo.size(); ++i) {
      if (rcs[i] != 0) return fail(h, rcs[i], "symbolic analysis failed: " + todo[i]->plan.error);
      if (todo[i]->plan.usize >= (int64_t)1 << 31) return fail(h, 1, "panel storage exceeds 2^31 entries per instance");
    }
  }
  for (Group* g : h->groups) {
    const pp::Plan& P = g->plan;
    GroupDev& d = g->dev;
    std::memset(&d, 0, sizeof(d));
    d.n = P.n; d.nc = g->nc_loc; d.batch = g->batch; d.bpad = (g->batch + WAVE - 1) / WAVE * WAVE;
    d.xs_row = 1; d.xs_lane = 0;
    d.rhsN = nullptr;
    d.nchunk = d.bpad / WAVE; d.npiv = P.npiv; d.nraw = g->nraw; d.usize = P.usize;
    int rc;
    std::vector<int> uoff(P.piv_uoff.begin(), P.piv_uoff.end());
    // widest column slice of a gather task per level (selects the kernel instantiation; the root front is gathered in
    // slices of PP_WMAX) and widest block pivot with ordinary scale tasks
    g->level_maxw.assign(P.n_levels, 1);
    for (int pp_ = 0; pp_ < P.npiv; ++pp_)
      g->level_maxw[P.piv_flevel[pp_]] = std::max(g->level_maxw[P.piv_flevel[pp_]], std::min(P.piv_w[pp_], PP_WMAX));
    {
      // chain fronts: header and panel records, LDS need per level
      std::vector<int> chdr, cpan, ccol, ccolN;
      g->chain_lds.assign((size_t)P.n_levels, 0);
      for (size_t c = 0; c < P.chain_m.size(); ++c) {
        // ([4]: first entry of the front's columns in chain_col / chain_colN (new column index / caller's row of it),
        // [5], [6]: the rows below the front's pivots = the rows of its last panel: offset into rowidx, count)
        {
          const int plast = P.chain_piv[(size_t)P.chain_ptr[c + 1] - 1];
          chdr.insert(chdr.end(), {P.chain_m[c], P.chain_w[c], P.chain_ptr[c + 1] - P.chain_ptr[c], P.chain_ptr[c], (int)ccol.size(),
                                   P.piv_rowptr[plast], P.piv_rowptr[plast + 1] - P.piv_rowptr[plast], 0});
          for (int t = P.chain_ptr[c]; t < P.chain_ptr[c + 1]; ++t)
            for (int q = 0; q < P.piv_w[P.chain_piv[t]]; ++q) {
              ccol.push_back(P.piv_start[P.chain_piv[t]] + q);
              ccolN.push_back(P.perm[P.piv_start[P.chain_piv[t]] + q]);
            }
        }
        for (int t = P.chain_ptr[c]; t < P.chain_ptr[c + 1]; ++t) {
          const int cp = P.chain_piv[t], cw = P.piv_w[cp];
          cpan.insert(cpan.end(), {cp, cw, (int)P.piv_uoff[cp], P.piv_boff[cp], P.piv_doff[cp], (int)P.piv_sub[cp], P.chain_col0[t],
                                   cw + (P.piv_rowptr[cp + 1] - P.piv_rowptr[cp])});
        }
        const size_t doubles = (size_t)P.chain_m[c] * ((size_t)(P.chain_w[c] | 1) + 4) + (size_t)P.chain_w[c] + 16;
        g->chain_lds[(size_t)P.chain_level[c]] = std::max(g->chain_lds[(size_t)P.chain_level[c]], doubles * sizeof(double));
      }
      d.chain_hdr = d.chain_pan = d.chain_col = nullptr;
      g->chain_colN = nullptr;
      if (!chdr.empty()) {
        if ((rc = dev_upload(h, g, &d.chain_hdr, chdr))) return rc;
        if ((rc = dev_upload(h, g, &d.chain_pan, cpan))) return rc;
        if ((rc = dev_upload(h, g, &d.chain_col, ccol))) return rc;
        if ((rc = dev_upload(h, g, &g->chain_colN, ccolN))) return rc;
      }
    }
    std::vector<int> ftask, stask, fdst_ptr, fent, srec;
    const double one = 1.0;
    int one_lo, one_hi;
    { int bits[2]; std::memcpy(bits, &one, sizeof(one)); one_lo = bits[0]; one_hi = bits[1]; }
    g->init_rec.clear();
    // raw entries that some canonical entry reads get a compact row in the transposed buffer; the rest
    // (typically the upper-triangle half) are never written
    std::vector<int> rawmap((size_t)std::max(g->nraw, 1), -1);
    // compact rows follow the raw order, so that a run of needed raw entries is a run of rows: the host boundary
    // uploads only those ([batch][nraw_used], pp_upload_values_compact) and the transposition needs no row map
    for (int v : g->can_idx) rawmap[v] = 0;
    g->used_raw.clear();
    for (int e = 0; e < g->nraw; ++e) if (rawmap[(size_t)e] == 0) { rawmap[(size_t)e] = (int)g->used_raw.size(); g->used_raw.push_back(e); }
    g->nraw_used = (int)g->used_raw.size();
    // expand the canonical initial-value entries into raw-value entries (duplicates are summed)
    fdst_ptr.reserve(P.fdst_ptr.size());
    fent.reserve(P.fentries.size() * 4 + 64);
    ftask.reserve(P.ftasks.size() * TASK_INTS);
    for (auto& t : P.ftasks) {
      const int nrow = t.r1 - t.r0;
      const int new_dptr0 = (int)fdst_ptr.size();
      // fourth field of a record: bits 0-7 the destination column of an initial value, or first column | columns << 4 of
      // a product entry; bits 8.. the number of
      // destination rows that END before this entry (0 inside a row; k_gather_flat closes that many rows first)
      int cur_row = 0;
      for (int dd = 0; dd < nrow; ++dd) {
        fdst_ptr.push_back((int)(fent.size() / 4));
        for (int e = P.fdst_ptr[t.dptr0 + dd]; e < P.fdst_ptr[t.dptr0 + dd + 1]; ++e) {
          const pp::FEntry& fe = P.fentries[e];
          if (fe.u >= 0) { fent.insert(fent.end(), {fe.u, fe.l, fe.wk, (fe.q & 255) | ((dd - cur_row) << 8)}); cur_row = dd; }
          else {
            // the L index of an initial-value record is a dummy (position 0, always valid)
            const int ce = -1 - fe.u;
            for (int q = g->can_ptr[ce]; q < g->can_ptr[ce + 1]; ++q)
            {
              g->init_rec.push_back((int)(fent.size() / 4));
              fent.insert(fent.end(), {-1 - rawmap[g->can_idx[q]], one_lo, one_hi, fe.q | ((dd - cur_row) << 8)});
              cur_row = dd;
            }
          }
        }
      }
      fdst_ptr.push_back((int)(fent.size() / 4));
      ftask.insert(ftask.end(), {t.piv, t.r0, t.r1, new_dptr0, t.kind, fdst_ptr[new_dptr0], (int)(fent.size() / 4),
                                 t.ws > 0 ? t.ws : P.piv_w[t.piv], (int)P.piv_uoff[t.piv], P.piv_boff[t.piv], P.piv_doff[t.piv],
                                 (int)P.piv_sub[t.piv], t.piece, t.npieces, P.piv_w[t.piv], t.qoff});
    }
    // tile tasks: their initial values as ordinary records in fent (the fused-source form rewrites them with the others),
    // their source-panel records with absent rows / columns pointing at the zero rows behind the panels
    // device form of the source-panel records: one STEP of 8 ints per source column {U position of row i x 4, L position of
    // column q x 4} (absent: the zero rows behind the panels), every task's steps padded to a multiple of PP_TILE_DEPTH with
    // all-zero steps -- a uniform, branch-free stream for the software pipeline of k_gather_tiles
    std::vector<int> ttask, trec;
    if (!P.ttasks.empty() && (uint64_t)(P.usize + PP_WMAX) * (uint64_t)d.nchunk >= ((uint64_t)1 << 32))
      return fail(h, 1, "tile tasks: panel storage x instance chunks exceeds 2^32 (PP_PLAN_TUNE chain_tiles=0 plans without them)");
    const int nck = d.nchunk;                 // operand rows in units of 64 doubles: position x chunks
    const int zpos = (int)P.usize;
    for (auto& t : P.ttasks) {
      const int s0 = (int)(trec.size() / 8);
      for (int r = t.te0; r < t.te1 && t.kind >= 0; ++r) {
        const int* rec = &P.trec[(size_t)r * PP_TREC_INTS];
        for (int k = 0; k < rec[0]; ++k)
          for (int q = 1; q <= 8; ++q) trec.push_back((int)((unsigned)(rec[q] < 0 ? zpos : rec[q] + k) * (unsigned)nck));
      }
      while ((trec.size() / 8 - (size_t)s0) % PP_TILE_DEPTH != 0)
        for (int q = 0; q < 8; ++q) trec.push_back((int)((unsigned)zpos * (unsigned)nck));
      const int s1 = (int)(trec.size() / 8);
      const int nrow = t.r1 - t.r0;
      const int e0 = (int)(fent.size() / 4);
      int cur_row = 0;
      for (int dd = 0; dd < nrow && t.kind >= 0; ++dd)
        for (int e = P.fdst_ptr[t.dptr0 + dd]; e < P.fdst_ptr[t.dptr0 + dd + 1]; ++e) {
          const pp::FEntry& fe = P.fentries[e];
          if (fe.u >= 0) continue;
          const int ce = -1 - fe.u;
          for (int q = g->can_ptr[ce]; q < g->can_ptr[ce + 1]; ++q) {
            g->init_rec.push_back((int)(fent.size() / 4));
            fent.insert(fent.end(), {-1 - rawmap[g->can_idx[q]], one_lo, one_hi, fe.q | ((dd - cur_row) << 8)});
            cur_row = dd;
          }
        }
      ttask.insert(ttask.end(), {t.piv, t.r0, t.r1, s0, t.kind, e0, (int)(fent.size() / 4), t.ws > 0 ? t.ws : P.piv_w[t.piv],
                                 (int)P.piv_uoff[t.piv], P.piv_boff[t.piv], s1, 0, t.piece, t.npieces, P.piv_w[t.piv], t.qoff});
    }
    for (int q = 0; q < 16 * PP_TILE_DEPTH; ++q) trec.push_back((int)((unsigned)zpos * (unsigned)nck));   // (the pipeline reads up to 2 PP_TILE_DEPTH - 2 steps ahead)
    d.ttask = d.trec = nullptr;
    if (!ttask.empty()) {
      if ((rc = dev_upload(h, g, &d.ttask, ttask))) return rc;
      if ((rc = dev_upload(h, g, &d.trec, trec))) return rc;
    }
    for (auto& t : P.stasks)
      stask.insert(stask.end(), {t.piv, t.r0, t.r1, -1, t.kind, 0, 0, P.piv_w[t.piv], (int)P.piv_uoff[t.piv],
                                 P.piv_boff[t.piv], P.piv_doff[t.piv], (int)P.piv_sub[t.piv], 0, 1, P.piv_w[t.piv], 0});
    {
      std::vector<int> wtask;
      for (auto& t : P.wtasks)
        wtask.insert(wtask.end(), {t.piv, t.r0, t.r1, -1, t.kind, 0, 0, P.piv_w[t.piv], (int)P.piv_uoff[t.piv],
                                   P.piv_boff[t.piv], P.piv_doff[t.piv], (int)P.piv_sub[t.piv], 0, 1, P.piv_w[t.piv], 0});
      g->wtask = nullptr;
      if (!wtask.empty() && (rc = dev_upload(h, g, &g->wtask, wtask))) return rc;
      g->front_inv = nullptr;
      if (P.front_piv >= 0 && (rc = dev_alloc<double>(h, g, &g->front_inv, (size_t)pp::PP_WF * pp::PP_WF * d.bpad))) return rc;
    }
    for (int q = 0; q < 16; ++q) fent.insert(fent.end(), {0, 0, 0, 0});   // slack for the vector record reads
    // tile records (one per panel) -> column-step records (one per panel column), with their own tile pointers
    std::vector<int> sptr(P.stile_ptr.size(), 0);
    for (size_t tix = 0; tix + 1 < P.stile_ptr.size(); ++tix) {
      sptr[tix] = (int)(srec.size() / 20);
      for (int ri = P.stile_ptr[tix]; ri < P.stile_ptr[tix + 1]; ++ri) {
        const auto& r = P.stile_rec[ri];
        const int w = P.piv_w[r.piv];
        for (int t = 0; t < w; ++t) {
          srec.insert(srec.end(), {w, (int)(P.piv_uoff[r.piv] + t), r.piv, t});
          for (int q = 0; q < 8; ++q) srec.push_back(r.slotA[q]);
          for (int q = 0; q < 8; ++q) srec.push_back(r.slotB[q]);
        }
      }
    }
    if (!sptr.empty()) sptr.back() = (int)(srec.size() / 20);
    if ((rc = dev_upload(h, g, &d.piv_w, P.piv_w))) return rc;
    if ((rc = dev_upload(h, g, &d.piv_start, P.piv_start))) return rc;
    if ((rc = dev_upload(h, g, &d.piv_uoff, uoff))) return rc;
    if ((rc = dev_upload(h, g, &d.piv_doff, P.piv_doff))) return rc;
    {
      std::vector<int> sub(P.piv_sub.begin(), P.piv_sub.end());
      if ((rc = dev_upload(h, g, &d.piv_sub, sub))) return rc;
      if ((rc = dev_upload(h, g, &d.piv_boff, P.piv_boff))) return rc;
      if ((rc = dev_upload(h, g, &d.piv_of_col, P.piv_of_col))) return rc;
    }
    if ((rc = dev_upload(h, g, &d.piv_rowptr, P.piv_rowptr))) return rc;
    if ((rc = dev_upload(h, g, &d.rowidx, P.rowidx))) return rc;
    if ((rc = dev_upload(h, g, &d.perm, P.perm))) return rc;
    if ((rc = dev_upload(h, g, &d.iperm, P.iperm))) return rc;
    if ((rc = dev_upload(h, g, &d.rawmap, rawmap))) return rc;
    if ((rc = ppi_build_residual_records(h, g, rawmap))) return rc;
    {
      std::vector<int> rtiles;
      for (int t0 = 0; t0 * 64 < g->nraw; ++t0) {
        bool any = false;
        for (int e = t0 * 64; e < std::min(g->nraw, t0 * 64 + 64) && !any; ++e) any = rawmap[(size_t)e] >= 0;
        if (any) rtiles.push_back(t0);
      }
      g->nraw_tiles = (int)rtiles.size();
      rtiles.push_back(0);
      if ((rc = dev_upload(h, g, &d.raw_tiles, rtiles))) return rc;
    }
    if ((rc = dev_upload(h, g, &d.ftask, ftask))) return rc;
    if ((rc = dev_upload(h, g, &d.stask, stask))) return rc;
    if ((rc = dev_upload(h, g, &d.fdst_ptr, fdst_ptr))) return rc;
    if ((rc = dev_upload(h, g, &d.fent, fent))) return rc;
    g->fent_host = fent;
    d.const_row = -1;
    if (!g->cmap_host.empty()) {      // [batch][nc_loc] -> [nc_loc][bpad] (padded lanes repeat instance 0: never used)
      std::vector<int> cm((size_t)std::max(g->nc_loc, 1) * d.bpad, 0);
      for (int c = 0; c < g->nc_loc; ++c)
        for (int b = 0; b < d.bpad; ++b)
          cm[(size_t)c * d.bpad + b] = g->cmap_host[(size_t)(b < g->batch ? b : 0) * g->nc_loc + c];
      if ((rc = dev_upload(h, g, &d.cmapT, cm))) return rc;
      d.xs_row = d.bpad; d.xs_lane = 1;
    }
    if ((rc = dev_upload(h, g, &d.clevel_col, P.clevel_col))) return rc;
    {
      std::vector<int> frec, brec;
      frec.reserve(P.clevel_col.size() * 4);
      brec.reserve(P.clevel_col.size() * 8);
      for (int c : P.clevel_col) {
        const int pv = P.piv_of_col[c], w = P.piv_w[pv], q = c - P.piv_start[pv];
        frec.insert(frec.end(), {c, P.perm[c], P.sfwd_eptr[c], P.sfwd_eptr[c + 1]});
        // (chain sweeps: a column of a chain front takes only the rows below the front's pivots here -- the last m - W rows
        // of its panel; the later columns of the front are k_chain_bwd's)
        int nr = P.piv_rowptr[pv + 1] - P.piv_rowptr[pv], j0 = 0;
        if (P.chain_sweeps_on && P.piv_chain[pv] >= 0) {
          const int fc = P.piv_chain[pv];
          j0 = nr - (P.chain_m[(size_t)fc] - P.chain_w[(size_t)fc]);
        }
        brec.insert(brec.end(), {c, w, q, nr - j0, P.piv_rowptr[pv] + j0,
                                 (int)(P.piv_uoff[pv] + (int64_t)(w + j0) * w + q), P.piv_doff[pv], P.piv_start[pv]});
      }
      // (chain sweeps: a column of a chain front always takes part in its level's launch -- its y is written to Y even
      // without contributions from outside the front, where k_chain_fwd finishes it)
      auto in_front = [&](int c) { return P.chain_sweeps_on && P.piv_chain[P.piv_of_col[c]] >= 0; };
      g->fwd_level_has_entries.assign((size_t)P.n_levels, 0);
      for (int l = 0; l < P.n_levels; ++l)
        for (int q = P.clevel_ptr[l]; q < P.clevel_ptr[l + 1]; ++q) {
          const int c = P.clevel_col[q];
          if (P.sfwd_eptr[c + 1] > P.sfwd_eptr[c] || in_front(c)) { g->fwd_level_has_entries[(size_t)l] = 1; break; }
        }
      // wave teams: a row / column with more than a couple of 16-entry load rounds is shared by 4 or 16 waves
      // (thresholds measured at C3: 16/48 and 16/64 were slower, 48/128 the same)
      auto team_of = [](int longest) { return longest > 96 ? 16 : longest > 32 ? 4 : 1; };
      g->fwd_level_team.assign((size_t)P.n_levels, 1);
      g->bwd_level_team.assign((size_t)P.n_levels, 1);
      g->fwd_level_maxrow.assign((size_t)P.n_levels, 0);
      g->bwd_level_maxrow.assign((size_t)P.n_levels, 0);
      for (int l = 0; l < P.n_levels; ++l) {
        int fmax = 0, bmax = 0;
        for (int q = P.clevel_ptr[l]; q < P.clevel_ptr[l + 1]; ++q) {
          const int c = P.clevel_col[q], pv = P.piv_of_col[c];
          fmax = std::max(fmax, P.sfwd_eptr[c + 1] - P.sfwd_eptr[c]);
          bmax = std::max(bmax, P.piv_rowptr[pv + 1] - P.piv_rowptr[pv]);
        }
        g->fwd_level_team[(size_t)l] = team_of(fmax);
        g->bwd_level_team[(size_t)l] = team_of(bmax);
        g->fwd_level_maxrow[(size_t)l] = fmax;
        g->bwd_level_maxrow[(size_t)l] = bmax;
      }
      if ((rc = dev_upload(h, g, &d.fwd_rec, frec))) return rc;
      if ((rc = dev_upload(h, g, &d.bwd_rec, brec))) return rc;
      // native-vector variants: a column without incoming entries keeps y = b, which then is read from the caller's
      // right-hand side (row perm[c]) instead of a copy; x is written and read in the caller's row order
      std::vector<uint8_t> noent((size_t)P.n, 0);
      for (int c = 0; c < P.n; ++c) noent[(size_t)c] = P.sfwd_eptr[c + 1] == P.sfwd_eptr[c] && !in_front(c);
      std::vector<int> zf(P.sfwd_zcol), zc2(P.crow_zcol), brn(brec), ro(P.rowidx);
      for (auto& z : zf) if (noent[(size_t)z]) z = -1 - P.perm[z];
      for (auto& z : zc2) if (noent[(size_t)z]) z = -1 - P.perm[z];
      for (int q = 0; q < 16; ++q) { zf.push_back(0); zc2.push_back(0); }
      for (size_t i = 0; i < brn.size(); i += 8) {
        const int p0 = brn[i + 7], w = brn[i + 1];
        brn[i] = P.perm[brn[i]];
        // y of the whole block pivot is read from the right-hand side only if NONE of its columns has incoming entries
        // (then its level was not launched in the forward sweep); a level that was launched has written y for all of
        // its columns.  (Columns of one block pivot may differ: a panel below may hold only some of them as rows.)
        bool none = true;
        for (int t = 0; t < w; ++t) none = none && noent[(size_t)(p0 + t)];
        if (none) brn[i + 7] = -1 - p0;
      }
      for (auto& r : ro) if (r < P.n) r = P.perm[r];
      if ((rc = dev_upload(h, g, &g->zcolN_f, zf))) return rc;
      if ((rc = dev_upload(h, g, &g->zcolN_c, zc2))) return rc;
      if ((rc = dev_upload(h, g, &g->brecN, brn))) return rc;
      if ((rc = dev_upload(h, g, &g->rowidx_o, ro))) return rc;
    }
    {
      std::vector<int> up(P.sfwd_upos), zc(P.sfwd_zcol), cu(P.crow_upos), cz(P.crow_zcol);
      for (int q = 0; q < 16; ++q) { up.push_back(0); zc.push_back(0); cu.push_back(0); cz.push_back(0); }
      if ((rc = dev_upload(h, g, &d.sfwd_eptr, P.sfwd_eptr))) return rc;
      if ((rc = dev_upload(h, g, &d.sfwd_upos, up))) return rc;
      if ((rc = dev_upload(h, g, &d.sfwd_zcol, zc))) return rc;
      if ((rc = dev_upload(h, g, &d.crow_eptr, P.crow_eptr))) return rc;
      if ((rc = dev_upload(h, g, &d.crow_upos, cu))) return rc;
      if ((rc = dev_upload(h, g, &d.crow_zcol, cz))) return rc;
    }
    if ((rc = dev_upload(h, g, &d.stile_a, P.stile_a))) return rc;
    if ((rc = dev_upload(h, g, &d.stile_b, P.stile_b))) return rc;
    if ((rc = dev_upload(h, g, &d.stile_ptr, sptr))) return rc;
    if ((rc = dev_upload(h, g, &d.stile_rec, srec))) return rc;
    g->ntiles = (int)P.stile_a.size();
    // 16 x 16 tiles for the MFMA form (k_schur_mfma): per (tile pair, panel column) one record with the positions of the
    // 16 + 16 rows; the records of a tile are cut into work items of at most PP_MT_SLICE records
    g->mt_wide = h->nc >= PP_MT_WIDE_NC && pp::env_switch("PP_NO_WIDE_SCHUR_TILES") == nullptr;
    if (g->mt_wide) {
      // 32 x 32 super-tiles (k_schur_mfma_wide): per (super-tile pair, panel column) one record with the positions of the
      // 32 + 32 rows; mt_a / mt_b per quarter (4 per super-tile, -1: the quarter above the diagonal), items {r0, r1, super, 0}
      std::map<std::pair<int, int>, std::vector<int>> by_super;
      for (int pv = 0; pv < P.npiv; ++pv) {
        const int w = P.piv_w[pv];
        std::vector<int> sl;
        std::vector<std::array<int, 32>> slots;
        for (int q = P.piv_rowptr[pv]; q < P.piv_rowptr[pv + 1]; ++q) {
          const int r = P.rowidx[(size_t)q];
          if (r < P.n) continue;
          const int c = r - P.n, si = c / 32;
          if (sl.empty() || sl.back() != si) { sl.push_back(si); std::array<int, 32> e; e.fill(-1); slots.push_back(e); }
          slots.back()[(size_t)(c % 32)] = w + (q - P.piv_rowptr[pv]);
        }
        for (size_t a = 0; a < sl.size(); ++a)
          for (size_t b2 = 0; b2 <= a; ++b2)
            for (int t = 0; t < w; ++t) {
              auto& v = by_super[{sl[a], sl[b2]}];
              for (int q = 0; q < 32; ++q) v.push_back(slots[a][(size_t)q] < 0 ? -1 : (int)(P.piv_uoff[pv] + (int64_t)slots[a][(size_t)q] * w + t));
              for (int q = 0; q < 32; ++q) v.push_back(slots[b2][(size_t)q] < 0 ? -1 : (int)(P.piv_uoff[pv] + (int64_t)slots[b2][(size_t)q] * w + t));
            }
      }
      std::vector<int> mta, mtb, mrec, mitem, mwptr{0};
      int super = 0;
      for (auto& kv : by_super) {
        for (int x = 0; x < 2; ++x)
          for (int y = 0; y < 2; ++y) {
            const int ta = 2 * kv.first.first + x, tb = 2 * kv.first.second + y;
            mta.push_back(ta >= tb ? ta : -1);
            mtb.push_back(ta >= tb ? tb : -1);
          }
        const int r0 = (int)(mrec.size() / 64);
        mrec.insert(mrec.end(), kv.second.begin(), kv.second.end());
        const int r1 = (int)(mrec.size() / 64);
        for (int r = r0; r < r1; r += PP_MT_SLICE) mitem.insert(mitem.end(), {r, std::min(r1, r + PP_MT_SLICE), super, 0});
        mwptr.push_back((int)(mitem.size() / 4));
        ++super;
      }
      g->nmt = (int)mta.size();                       // quarters (workgroups of the reduction)
      g->nmt_items = (int)(mitem.size() / 4);
      if ((rc = dev_upload(h, g, &d.mt_a, mta))) return rc;
      if ((rc = dev_upload(h, g, &d.mt_b, mtb))) return rc;
      if ((rc = dev_upload(h, g, &d.mt_rec, mrec))) return rc;
      if ((rc = dev_upload(h, g, &d.mt_item, mitem))) return rc;
      if ((rc = dev_upload(h, g, &d.mt_wptr, mwptr))) return rc;
    } else {
      std::map<std::pair<int, int>, std::vector<int>> by_tile;      // (ta, tb) -> records of 32 ints
      for (int pv = 0; pv < P.npiv; ++pv) {
        const int w = P.piv_w[pv];
        std::vector<int> tl;
        std::vector<std::array<int, 16>> slots;
        for (int q = P.piv_rowptr[pv]; q < P.piv_rowptr[pv + 1]; ++q) {
          const int r = P.rowidx[(size_t)q];
          if (r < P.n) continue;
          const int c = r - P.n, ti = c / 16;
          if (tl.empty() || tl.back() != ti) { tl.push_back(ti); std::array<int, 16> e; e.fill(-1); slots.push_back(e); }
          slots.back()[(size_t)(c % 16)] = w + (q - P.piv_rowptr[pv]);      // row slot inside the panel
        }
        for (size_t a = 0; a < tl.size(); ++a)
          for (size_t b2 = 0; b2 <= a; ++b2)
            for (int t = 0; t < w; ++t) {
              auto& v = by_tile[{tl[a], tl[b2]}];
              for (int q = 0; q < 16; ++q) v.push_back(slots[a][(size_t)q] < 0 ? -1 : (int)(P.piv_uoff[pv] + (int64_t)slots[a][(size_t)q] * w + t));
              for (int q = 0; q < 16; ++q) v.push_back(slots[b2][(size_t)q] < 0 ? -1 : (int)(P.piv_uoff[pv] + (int64_t)slots[b2][(size_t)q] * w + t));
            }
      }
      std::vector<int> mta, mtb, mrec, mitem, mwptr{0};
      for (auto& kv : by_tile) {
        mta.push_back(kv.first.first); mtb.push_back(kv.first.second);
        const int r0 = (int)(mrec.size() / 32);
        mrec.insert(mrec.end(), kv.second.begin(), kv.second.end());
        const int r1 = (int)(mrec.size() / 32);
        for (int r = r0; r < r1; r += PP_MT_SLICE) { mitem.push_back(r); mitem.push_back(std::min(r1, r + PP_MT_SLICE)); }
        mwptr.push_back((int)(mitem.size() / 2));
      }
      g->nmt = (int)mta.size();
      g->nmt_items = (int)(mitem.size() / 2);
      if ((rc = dev_upload(h, g, &d.mt_a, mta))) return rc;
      if ((rc = dev_upload(h, g, &d.mt_b, mtb))) return rc;
      if ((rc = dev_upload(h, g, &d.mt_rec, mrec))) return rc;
      if ((rc = dev_upload(h, g, &d.mt_item, mitem))) return rc;
      if ((rc = dev_upload(h, g, &d.mt_wptr, mwptr))) return rc;
    }
  }
  int rc;
  if (h->btd && (h->G < 1 || h->gs < 1 || h->gs > 512 || (int64_t)h->G * h->gs != nc))
    return fail(h, 3, "block-tridiagonal coupling structure: need n_c = G * gs with 1 <= gs <= 512");
  for (Group* g : h->groups)
    if (h->btd && g->cmap_host.empty() && nc > 0) return fail(h, 3, "block-tridiagonal S needs mapped groups (pp_add_group_mapped)");
  const size_t nn = schur_doubles(h);
  const size_t nd = h->btd ? 1 : nn;            // the dense factor copies are not needed for a block-tridiagonal S
  if ((rc = dev_alloc<double>(h, nullptr, &h->S_own, nn + PP_TAIL))) return rc;
  if ((rc = dev_alloc<double>(h, nullptr, &h->Sfac, nd))) return rc;
  if ((rc = dev_alloc<double>(h, nullptr, &h->Sldl, nd))) return rc;
  if ((rc = dev_alloc<int>(h, nullptr, &h->scatter_err, 4))) return rc;
  PP_HIP(hipMemset(h->scatter_err, 0, 4 * sizeof(int)));
  if (h->btd) {
    const size_t g2 = (size_t)h->gs * h->gs;
    if ((rc = dev_alloc<double>(h, nullptr, &h->btd_fac, nn))) return rc;
    if ((rc = dev_alloc<double>(h, nullptr, &h->btd_inv, (size_t)h->G * g2))) return rc;
    if ((rc = dev_alloc<double>(h, nullptr, &h->btd_klo, (size_t)h->G * g2))) return rc;
    if ((rc = dev_alloc<double>(h, nullptr, &h->btd_kup, (size_t)h->G * g2))) return rc;
    if ((rc = dev_alloc<double>(h, nullptr, &h->btd_ylo, (size_t)h->G * g2))) return rc;
    if ((rc = dev_alloc<double>(h, nullptr, &h->btd_yup, (size_t)h->G * g2))) return rc;
    if ((rc = dev_alloc<double>(h, nullptr, &h->btd_vec, 8 * (size_t)nc + 64))) return rc;    // BK work | b | u
    if ((rc = dev_alloc<int>(h, nullptr, &h->btd_ipiv, nc))) return rc;
    if ((rc = dev_alloc<int>(h, nullptr, &h->btd_info, 4 * (size_t)h->G))) return rc;
    if ((rc = build_btd_schedule(h))) return rc;
  }
  if ((rc = dev_alloc<double>(h, nullptr, &h->dvec, nc))) return rc;
  if ((rc = dev_alloc<int>(h, nullptr, &h->dense_mode, 4))) return rc;
  PP_HIP(hipMemset(h->dense_mode, 0, 4 * sizeof(int)));
  // Q: dense n_c x n_c; for a block-tridiagonal S (the layout of S, tens of MB) only when a caller hands over a flat Q --
  // the sparse form of pp_factor_schur_corner needs none
  if (!h->btd && (rc = dev_alloc<double>(h, nullptr, &h->Qd, nn))) return rc;
  if ((rc = dev_alloc<double>(h, nullptr, &h->work, 2 * (size_t)nc))) return rc;
  if ((rc = dev_alloc<double>(h, nullptr, &h->rs_own, nc))) return rc;
  if ((rc = dev_alloc<double>(h, nullptr, &h->rcd, nc))) return rc;
  if ((rc = dev_alloc<double>(h, nullptr, &h->xc, nc))) return rc;
  if ((rc = dev_alloc<int>(h, nullptr, &h->ipiv, nc))) return rc;
  if ((rc = dev_alloc<int>(h, nullptr, &h->bkinfo, 4))) return rc;
  if ((rc = dev_alloc<int>(h, nullptr, &h->counters, 4 * PP_CSLOTS))) return rc;
  PP_HIP(hipMemset(h->counters, 0, 4 * PP_CSLOTS * sizeof(int)));     // (afterwards cleared by the kernel that writes the tail)
  {
    void* hp = nullptr;
    void* dp = nullptr;
    PP_HIP(hipHostMalloc(&hp, 8 * sizeof(long long), hipHostMallocMapped));
    std::memset(hp, 0, 8 * sizeof(long long));
    PP_HIP(hipHostGetDevicePointer(&dp, hp, 0));
    h->status_host = (volatile long long*)hp;
    h->status_dev = (long long*)dp;
    h->status_seq = 0;
  }
  // stream of the dense phase (dense.hip): made here, not in the first factorisation -- and only for a dense S: the
  // runtime maps streams onto a few hardware queues, and one more stream made the three group streams of a time-staged
  // problem share a queue (C4: 7.9 -> 11.2 ms per step)
  if (!h->dense_stream && h->dense_overlap && !h->btd && nc > 0 && h->groups.size() <= 2) {
    PP_HIP(hipStreamCreateWithFlags(&h->dense_stream, hipStreamNonBlocking));
    PP_HIP(hipEventCreateWithFlags(&h->ev_dense_fork, hipEventDisableTiming));
    PP_HIP(hipEventCreateWithFlags(&h->ev_dense_done, hipEventDisableTiming));
  }
  h->S = h->S_own;
  h->rs = h->rs_own;
  PP_HIP(hipMemset(h->S, 0, (nn + PP_TAIL) * sizeof(double)));
  PP_HIP(hipMemset(h->rs, 0, std::max<size_t>(nc, 1) * sizeof(double)));
  PP_HIP(hipMemset(h->bkinfo, 0, 4 * sizeof(int)));
  h->symbolic_done = true;
  // value storage (factor panels, work vectors): sized by the plan; if it does not fit the handle's budget the
  // symbolic phase still succeeds (the plan is valid) and the numeric phase reports not_enough_memory until
  // increase_memory_allocation has raised the budget (reference: ma27_interface.py:126-131, 153-154)
  h->values_allocated = false;
  h->mem_required = value_storage_bytes(h);
  (void)alloc_value_storage(h);
  h->err.clear();
  return 0;
}

int pp_upload_values(pp_handle h, int group, const double* raw, int on_device) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_upload_values: bad group or symbolic phase not finished");
  PP_HIP(hipSetDevice(h->device));
  if (int rc = alloc_value_storage(h)) return rc;
  if (!g->dev.raw) { if (int rc = ensure_optional(h, g, OPT_RAW)) return rc; }
  g->input_mode = Group::IN_RAW;
  if (g->dev.raw == g->raw_own) invalidate_stage_mirror(g);     // ([batch][nraw] over the buffer the compact rows mirror)
  const size_t bytes = (size_t)g->batch * g->nraw * sizeof(double);
  if (bytes == 0 || raw == g->dev.raw) return 0;
  PP_HIP(hipMemcpyAsync(g->dev.raw, raw, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
  return 0;
}

int pp_upload_values_compact(pp_handle h, int group, const double* compact, int row0, int nrows, int on_device) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_upload_values_compact: bad group or symbolic phase not finished");
  if (row0 < 0 || nrows < 0 || row0 + nrows > g->batch) return fail(h, 3, "pp_upload_values_compact: row range outside the batch");
  PP_HIP(hipSetDevice(h->device));
  if (int rc = ensure_optional(h, g, OPT_RAW)) return rc;
  g->input_mode = Group::IN_COMPACT;
  const size_t stride = (size_t)g->nraw_used;
  if (nrows == 0 || stride == 0) return 0;
  PP_HIP(hipMemcpyAsync(g->raw_own + (size_t)row0 * stride, compact + (size_t)row0 * stride, (size_t)nrows * stride * sizeof(double),
                        on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
  // (rows sent from the staging array the threaded paths use mirror it from now on; anything else breaks the mirror)
  if (g->staged_row_valid.size() != (size_t)g->batch) g->staged_row_valid.assign((size_t)g->batch, 0);
  const uint8_t ok = (!on_device && compact == g->stage_host) ? 1 : 0;
  for (int r = row0; r < row0 + nrows; ++r) g->staged_row_valid[(size_t)r] = ok;
  return 0;
}

int pp_used_raw_entries(pp_handle h, int group, int32_t* out, int capacity) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_used_raw_entries: bad group or symbolic phase not finished");
  if (capacity < (int)g->used_raw.size()) return fail(h, 3, "pp_used_raw_entries: buffer too small");
  std::memcpy(out, g->used_raw.data(), g->used_raw.size() * sizeof(int));
  return 0;
}

int pp_set_value_map(pp_handle h, int group, int nsrc, const int32_t* src_of_raw, const double* coef_of_raw) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_set_value_map: bad group or symbolic phase not finished");
  if (nsrc < 0 || !src_of_raw || !coef_of_raw) return fail(h, 3, "pp_set_value_map: bad arguments");
  PP_HIP(hipSetDevice(h->device));
  PP_HIP(hipStreamSynchronize(h->stream));
  std::vector<int> ms(g->used_raw.size() + 1, -1);
  std::vector<double> mc(g->used_raw.size() + 1, 0.0);
  for (size_t j = 0; j < g->used_raw.size(); ++j) {
    const int e = g->used_raw[j];
    if (src_of_raw[e] >= nsrc) return fail(h, 3, "pp_set_value_map: source row out of range");
    ms[j] = src_of_raw[e] < 0 ? -1 : src_of_raw[e];
    mc[j] = coef_of_raw[e];
  }
  for (void* p : {(void*)g->map_src, (void*)g->map_coef, (void*)g->src_own}) if (p) (void)hipFree(p);
  g->map_src = nullptr; g->map_coef = nullptr; g->src_own = nullptr; g->src = nullptr;
  g->nsrc = nsrc;
  int rc;
  if ((rc = dev_alloc(h, (Group*)nullptr, &g->map_src, ms.size()))) return rc;
  if ((rc = dev_alloc(h, (Group*)nullptr, &g->map_coef, mc.size()))) return rc;
  PP_HIP(hipMemcpy(g->map_src, ms.data(), ms.size() * sizeof(int), hipMemcpyHostToDevice));
  PP_HIP(hipMemcpy(g->map_coef, mc.data(), mc.size() * sizeof(double), hipMemcpyHostToDevice));
  // entry records for the fused path: an initial-value record reads its source row directly (row nsrc = the constant
  // 1) and carries its coefficient, so the factorisation kernels gather from the sources themselves
  {
    std::vector<int> fs(g->fent_host);
    for (int pos : g->init_rec) {
      const int row = -1 - fs[(size_t)4 * pos];            // compact row of the transposed input
      const int sidx = ms[(size_t)row];
      const double c = mc[(size_t)row];
      int bits[2];
      std::memcpy(bits, &c, sizeof(c));
      fs[(size_t)4 * pos] = -1 - (sidx >= 0 ? sidx : nsrc);
      fs[(size_t)4 * pos + 1] = bits[0];
      fs[(size_t)4 * pos + 2] = bits[1];
    }
    if (g->fent_src) { (void)hipFree(g->fent_src); g->fent_src = nullptr; }
    if ((rc = dev_alloc(h, (Group*)nullptr, &g->fent_src, fs.size()))) return rc;
    PP_HIP(hipMemcpy(g->fent_src, fs.data(), fs.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  return ppi_residual_value_map(h, g, ms, mc);
}

double* pp_source_buffer(pp_handle h, int group) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done || !g->map_src) return nullptr;
  if (!g->src_own) {
    if (dev_alloc(h, (Group*)nullptr, &g->src_own, (size_t)std::max(g->nsrc, 1) * (size_t)g->dev.bpad)) return nullptr;
    if (hipMemset(g->src_own, 0, (size_t)std::max(g->nsrc, 1) * (size_t)g->dev.bpad * sizeof(double)) != hipSuccess) return nullptr;
  }
  if (!g->src) g->src = g->src_own;
  return g->src_own;
}

int pp_bind_source_buffer(pp_handle h, int group, double* dev_ptr) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done || !g->map_src) return fail(h, 3, "pp_bind_source_buffer: bad group or no value map");
  if (!dev_ptr && !pp_source_buffer(h, group)) return fail(h, 1, "pp_bind_source_buffer: could not allocate the source buffer");
  g->src = dev_ptr ? dev_ptr : g->src_own;
  g->input_mode = Group::IN_SOURCES;
  return 0;
}

int pp_upload_sources(pp_handle h, int group, const double* src, int on_device) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done || !g->map_src) return fail(h, 3, "pp_upload_sources: bad group or no value map");
  PP_HIP(hipSetDevice(h->device));
  if (int rc = ensure_optional(h, g, OPT_RAW)) return rc;       // (host sources are staged through the raw buffer)
  if (!pp_source_buffer(h, group)) return fail(h, 1, "pp_upload_sources: could not allocate the source buffer");
  // [batch][nsrc] (one row per block, the producer's natural layout on the host) -> [nsrc][bpad]: staged through the
  // raw buffer (nsrc <= nraw is not required: the copy is done in slabs of whole rows)
  const size_t per = (size_t)g->nsrc;
  if (per == 0) { g->src = g->src_own; g->input_mode = Group::IN_SOURCES; return 0; }
  if (per > (size_t)g->nraw) return fail(h, 3, "pp_upload_sources: more sources than raw entries per block");
  invalidate_stage_mirror(g);                                    // (staged through raw_own)
  PP_HIP(hipMemcpyAsync(g->raw_own, src, (size_t)g->batch * per * sizeof(double),
                        on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_transpose_in, dim3((unsigned)((g->nsrc + 63) / 64) * g->dev.nchunk), dim3(256), 0, h->stream, g->raw_own,
                     g->src_own, (const int*)nullptr, g->batch, g->nsrc, g->dev.bpad, 1, (const int*)nullptr);
  PP_HIP(hipGetLastError());
  g->src = g->src_own;
  g->input_mode = Group::IN_SOURCES;
  return 0;
}

double* pp_raw_buffer(pp_handle h, int group) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done || alloc_value_storage(h)) return nullptr;
  if (!g->dev.raw && ensure_optional(h, g, OPT_RAW)) return nullptr;
  if (g->dev.raw == g->raw_own) invalidate_stage_mirror(g);     // (the caller writes into it)
  return g->dev.raw;
}

int pp_upload_rhs(pp_handle h, int group, const double* rhs, int on_device) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_upload_rhs: bad group or symbolic phase not finished");
  PP_HIP(hipSetDevice(h->device));
  if (int rc = alloc_value_storage(h)) return rc;
  if (!g->dev.rhs) { if (int rc = ensure_optional(h, g, OPT_RHS)) return rc; }
  const size_t bytes = (size_t)g->batch * g->plan.n * sizeof(double);
  if (rhs == g->dev.rhs) return 0;
  PP_HIP(hipMemcpyAsync(g->dev.rhs, rhs, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
  return 0;
}

double* pp_rhs_buffer(pp_handle h, int group) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done || alloc_value_storage(h)) return nullptr;
  if (!g->dev.rhs && ensure_optional(h, g, OPT_RHS)) return nullptr;
  return g->dev.rhs;
}

int pp_bind_solution_buffer(pp_handle h, int group, double* dev_ptr) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_bind_solution_buffer: bad group");
  if (int rc = alloc_value_storage(h)) return rc;
  g->dev.xout = dev_ptr ? dev_ptr : g->xout_own;
  return 0;
}

int pp_bind_native_vectors(pp_handle h, int group, const double* rhs_dev, double* x_dev) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_bind_native_vectors: bad group");
  // (the right-hand side alone: enough for pp_solve_forward -- a forward sweep enqueued ahead of the back-solve)
  if (rhs_dev == nullptr && x_dev != nullptr) return fail(h, 3, "pp_bind_native_vectors: a solution buffer needs the right-hand side");
  if (int rc = alloc_value_storage(h)) return rc;
  g->rhs_native = rhs_dev;
  g->x_native = x_dev;
  return 0;
}

int pp_download_solution(pp_handle h, int group, double* x, int on_device) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_download_solution: bad group");
  PP_HIP(hipSetDevice(h->device));
  const size_t bytes = (size_t)g->batch * g->plan.n * sizeof(double);
  if (!g->dev.xout) return fail(h, 3, "pp_download_solution: no solution in the [instance][row] layout (native vectors bound?)");
  if (x != g->dev.xout)
    PP_HIP(hipMemcpyAsync(x, g->dev.xout, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
  if (!on_device) PP_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

double* pp_solution_buffer(pp_handle h, int group) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done || alloc_value_storage(h)) return nullptr;
  if (!g->dev.xout && ensure_optional(h, g, OPT_XOUT)) return nullptr;
  return g->dev.xout;
}

int pp_bind_raw_buffer(pp_handle h, int group, double* dev_ptr) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_bind_raw_buffer: bad group");
  g->dev.raw = dev_ptr ? dev_ptr : g->raw_own;
  g->input_mode = Group::IN_RAW;
  return 0;
}

int pp_bind_rhs_buffer(pp_handle h, int group, double* dev_ptr) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_bind_rhs_buffer: bad group");
  g->dev.rhs = dev_ptr ? dev_ptr : g->rhs_own;
  return 0;
}

int pp_set_supernodes(pp_handle h, int wmax, int tol_rows) {
  if (!h) return 3;
  if (wmax < 0 || wmax > PP_WMAX) return fail(h, 3, "supernode width must be 0 (default) .. PP_WMAX");
  h->sn_wmax = wmax;
  h->sn_tol = tol_rows;
  return 0;
}

int pp_set_instance_splits(pp_handle h, int nsplit) {
  if (!h) return 3;
  if (nsplit < 0 || nsplit > PP_MAX_SPLIT) return fail(h, 3, "instance splits must be 0 (automatic) .. 8");
  h->nsplit_req = nsplit;
  return 0;
}

int pp_set_dense_policy(pp_handle h, int policy) {
  if (!h) return 3;
  if (policy != 0 && policy != 1) return fail(h, 3, "dense policy must be 0 (auto) or 1 (Bunch-Kaufman only)");
  h->dense_policy = policy;
  return 0;
}

int pp_get_dense_mode(pp_handle h, int* mode_out) {
  if (!h || !h->schur_done) return fail(h, 3, "pp_get_dense_mode before pp_factor_schur");
  PP_HIP(hipSetDevice(h->device));
  if (int rc = join_dense(h)) return rc;
  PP_HIP(hipMemcpyAsync(mode_out, h->dense_mode, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  PP_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

int pp_profile(pp_handle h, int enable) {
  if (!h) return 3;
  h->profile = enable != 0;
  for (int i = 0; i < PP_NPHASE; ++i) {
    h->phase_ms[i] = 0.0; h->phase_launches[i] = 0; h->phase_calls[i] = 0; h->ev_used[i] = false;
  }
  return 0;
}

int pp_phase_times(pp_handle h, double ms_out[8], int32_t launches_out[8], int32_t calls_out[8]) {
  if (!h) return 3;
  PP_HIP(hipSetDevice(h->device));
  PP_HIP(hipStreamSynchronize(h->stream));
  for (int i = 0; i < PP_NPHASE; ++i) {
    if (h->ev_used[i]) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, h->ev[i][0], h->ev[i][1]) == hipSuccess) h->phase_ms[i] += ms;
      h->ev_used[i] = false;
    }
    if (i >= PP_NPHASE_SOLVER) continue;
    ms_out[i] = h->phase_ms[i];
    launches_out[i] = h->phase_launches[i];
    calls_out[i] = h->phase_calls[i];
  }
  return 0;
}

int pp_ip_phase_times(pp_handle h, double ms_out[4], int32_t launches_out[4], int32_t calls_out[4]) {
  if (!h) return 3;
  double ms[8]; int32_t l8[8], c8[8];
  if (int rc = pp_phase_times(h, ms, l8, c8)) return rc;       // (harvests all brackets)
  for (int i = 0; i < 4; ++i) {
    ms_out[i] = h->phase_ms[PP_NPHASE_SOLVER + i];
    launches_out[i] = h->phase_launches[PP_NPHASE_SOLVER + i];
    calls_out[i] = h->phase_calls[PP_NPHASE_SOLVER + i];
  }
  return 0;
}

int pp_increase_memory_allocation(pp_handle h, double factor) {
  if (!h) return 3;
  if (!(factor > 0.0)) return fail(h, 3, "memory allocation factor must be positive");
  h->mem_factor *= factor;
  return 0;
}

int pp_set_memory_budget(pp_handle h, int64_t bytes) {
  if (!h) return 3;
  if (bytes < 0) return fail(h, 3, "memory budget must be >= 0 (0: no limit)");
  h->mem_budget = bytes;
  h->mem_factor = 1.0;
  return 0;
}

int pp_memory_info(pp_handle h, int64_t out[3]) {
  if (!h) return 3;
  out[0] = h->mem_required;
  out[1] = h->mem_budget > 0 ? (int64_t)((double)h->mem_budget * h->mem_factor) : 0;
  int64_t allocated = 0;      // what is allocated now: the optional input / output copies only once something used them
  if (h->values_allocated) {
    allocated = h->mem_required;
    for (Group* g : h->groups) {
      const int64_t bp = g->dev.bpad;
      if (!g->raw_own) allocated -= 8 * (int64_t)g->batch * g->nraw;
      if (!g->rawT_own) allocated -= 8 * (int64_t)std::max(g->nraw_used, 1) * bp;
      if (!g->rhs_own) allocated -= 8 * (int64_t)g->batch * g->plan.n;
      if (!g->xout_own) allocated -= 8 * (int64_t)g->batch * g->plan.n;
      if (!g->dev.X) allocated -= 8 * (int64_t)g->plan.n * bp;
    }
  }
  out[2] = allocated;
  return 0;
}

int pp_bcr_block_paths(pp_handle h, int32_t out[2]) {
  if (!h) return 3;
  out[0] = out[1] = 0;
  if (!h->btd || !h->schur_done || !h->btd_info) return 0;
  PP_HIP(hipSetDevice(h->device));
  std::vector<int> info(4 * (size_t)h->G);
  PP_HIP(hipMemcpyAsync(info.data(), h->btd_info, info.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  PP_HIP(hipStreamSynchronize(h->stream));
  for (int t = 0; t < h->G; ++t) out[info[4 * (size_t)t + 3] == 1 ? 0 : 1] += 1;
  return 0;
}

int pp_synchronize(pp_handle h) {
  if (!h) return 3;
  PP_HIP(hipSetDevice(h->device));
  if (int rc = join_dense(h)) return rc;
  PP_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

int pp_group_stats(pp_handle h, int group, int64_t out[16]) {
  Group* g = get_group(h, group);
  if (!g) return fail(h, 3, "pp_group_stats: bad group");
  const pp::Plan& P = g->plan;
  const int64_t v[16] = {P.n, P.nc, g->batch, P.npiv, P.n_2x2, P.n_levels, P.nnz_L, P.usize, P.flops_factor,
                         P.flops_schur, (int64_t)P.ftasks.size(), (int64_t)P.fentries.size(), (int64_t)P.stile_a.size(),
                         (int64_t)P.stile_rec.size(), P.ncan, g->nraw};
  std::memcpy(out, v, sizeof(v));
  return 0;
}

int pp_group_stats_ex(pp_handle h, int group, int64_t out[16]) {
  Group* g = get_group(h, group);
  if (!g) return fail(h, 3, "pp_group_stats_ex: bad group");
  const pp::Plan& P = g->plan;
  int64_t coupling_entries = 0;
  for (int p = 0; p < P.npiv; ++p) coupling_entries += (int64_t)P.piv_ncrow[p] * P.piv_w[p];
  int64_t launches_factor = 0, launches_fwd = 1, launches_bwd = 1;
  for (int l = 0; l < P.n_levels; ++l) {
    launches_factor += (P.flevel_ptr[l + 1] > P.flevel_ptr[l]) + (P.slevel_ptr[l + 1] > P.slevel_ptr[l]) +
                       ((P.front_piv >= 0 && P.piv_flevel[P.front_piv] == l) ? 1 + (P.wtasks.empty() ? 0 : 1) : 0) +
                       (P.chain_lvl_ptr[l + 1] > P.chain_lvl_ptr[l]) + (P.tlevel_ptr[l + 1] > P.tlevel_ptr[l]);
    if (l < (int)g->fwd_level_has_entries.size() && g->fwd_level_has_entries[(size_t)l]) ++launches_fwd;
    if (P.clevel_ptr[l + 1] > P.clevel_ptr[l]) ++launches_bwd;
  }
  const int64_t index_bytes = 4 * ((int64_t)P.fentries.size() * 4 + (int64_t)P.ftasks.size() * TASK_INTS +
                                   (int64_t)P.stasks.size() * TASK_INTS + (int64_t)P.fdst_ptr.size() +
                                   2 * (int64_t)P.sfwd_upos.size() + 2 * (int64_t)P.crow_upos.size() + (int64_t)P.rowidx.size() +
                                   12 * (int64_t)P.n + 20 * (int64_t)P.stile_rec.size() * 4);
  const int64_t v[16] = {g->nraw_used, P.dsize, P.bsize, coupling_entries, index_bytes, (int64_t)P.sfwd_upos.size(),
                         (int64_t)P.crow_upos.size(), g->nsrc, launches_factor, launches_fwd, launches_bwd,
                         (int64_t)g->dev.bpad, (int64_t)g->dev.nchunk, (int64_t)g->ntiles, (int64_t)P.tail_level0, 0};
  std::memcpy(out, v, sizeof(v));
  return 0;
}

int pp_group_perm(pp_handle h, int group, int32_t* perm) {
  Group* g = get_group(h, group);
  if (!g) return fail(h, 3, "pp_group_perm: bad group");
  std::memcpy(perm, g->plan.perm.data(), sizeof(int) * g->plan.n);
  return 0;
}

int pp_set_diagonal_classes(pp_handle h, int group, const int8_t* cls) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_set_diagonal_classes: bad group or symbolic phase not finished");
  PP_HIP(hipSetDevice(h->device));
  std::vector<int> rows, kinds;
  for (int i = 0; i < g->plan.n; ++i) {
    if (cls[i] == 0) continue;
    if (cls[i] != 1 && cls[i] != 2) return fail(h, 3, "pp_set_diagonal_classes: classes are 0, 1 (Hessian) or 2 (constraint)");
    const int ce = g->diag_can[(size_t)i];
    if (ce < 0)
      return fail(h, 3, "pp_set_diagonal_classes: row " + std::to_string(i) +
                            " has a class but no diagonal entry in the planned pattern");
    // the shift goes to the first raw duplicate of the canonical diagonal entry
    const int raw = g->can_idx[(size_t)g->can_ptr[(size_t)ce]];
    rows.push_back(-1 - raw);
    kinds.push_back((int)cls[i]);
  }
  // raw index -> compact row of the transposed input (same rule as pp_end_symbolic)
  {
    std::vector<int> rawmap((size_t)std::max(g->nraw, 1), -1);
    for (size_t j = 0; j < g->used_raw.size(); ++j) rawmap[(size_t)g->used_raw[j]] = (int)j;
    for (auto& r : rows) r = rawmap[(size_t)(-1 - r)];
  }
  PP_HIP(hipStreamSynchronize(h->stream));      // a previous shifted factorisation may still read the old arrays
  if (g->shift_row) { (void)hipFree(g->shift_row); g->shift_row = nullptr; }
  if (g->shift_cls) { (void)hipFree(g->shift_cls); g->shift_cls = nullptr; }
  g->nshift = 0;
  rows.push_back(0); kinds.push_back(0);
  int rc;
  if ((rc = dev_alloc(h, (Group*)nullptr, &g->shift_row, rows.size()))) return rc;
  if ((rc = dev_alloc(h, (Group*)nullptr, &g->shift_cls, kinds.size()))) return rc;
  PP_HIP(hipMemcpy(g->shift_row, rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice));
  PP_HIP(hipMemcpy(g->shift_cls, kinds.data(), kinds.size() * sizeof(int), hipMemcpyHostToDevice));
  g->nshift = (int)rows.size() - 1;
  return 0;
}

// Host-side staging (no device work): for every block whose raw COO index arrays equal the group's reference
// arrays, the values go to the block's row of the staging array; same_out[i] tells the caller which blocks it has to
// canonicalise itself (quirk Q7).  Compare + copy are memory-bound, so they are spread over host threads.
// runs (may be null = everything): triples {first entry, length, destination offset in the row} of the K data and of
// the border data that are copied -- the entries some canonical entry reads (a KKT block handed over with both
// triangles has whole runs of upper-triangle entries nobody reads: they are neither staged nor uploaded).
namespace {
struct StageArgs {
  const int32_t* const* kr; const int32_t* const* kc; const double* const* kd; const int64_t* knnz;
  const int32_t* const* br; const int32_t* const* bc; const double* const* bd; const int64_t* bnnz;
  const int32_t *ref_kr, *ref_kc; int64_t ref_knnz; const int32_t *ref_br, *ref_bc; int64_t ref_bnnz;
  int nrunsK; const int64_t* runsK; int nrunsB; const int64_t* runsB;
  double* staging; int64_t row_stride; const int32_t* slots; uint8_t* same_out;
};
void stage_range(const StageArgs& a, int i0, int i1) {
  for (int i = i0; i < i1; ++i) {
    bool same = a.knnz[i] == a.ref_knnz && a.bnnz[i] == a.ref_bnnz;
    const size_t kb = (size_t)a.ref_knnz * sizeof(int32_t), bb = (size_t)a.ref_bnnz * sizeof(int32_t);
    same = same && (a.kr[i] == a.ref_kr || kb == 0 || std::memcmp(a.kr[i], a.ref_kr, kb) == 0);
    same = same && (a.kc[i] == a.ref_kc || kb == 0 || std::memcmp(a.kc[i], a.ref_kc, kb) == 0);
    same = same && (a.br[i] == a.ref_br || bb == 0 || std::memcmp(a.br[i], a.ref_br, bb) == 0);
    same = same && (a.bc[i] == a.ref_bc || bb == 0 || std::memcmp(a.bc[i], a.ref_bc, bb) == 0);
    if (same) {
      double* row = a.staging + (size_t)a.slots[i] * (size_t)a.row_stride;
      if (!a.runsK) {
        if (a.ref_knnz > 0) std::memcpy(row, a.kd[i], (size_t)a.ref_knnz * sizeof(double));
        if (a.ref_bnnz > 0) std::memcpy(row + a.ref_knnz, a.bd[i], (size_t)a.ref_bnnz * sizeof(double));
      } else {
        for (int r = 0; r < a.nrunsK; ++r)
          std::memcpy(row + a.runsK[3 * r + 2], a.kd[i] + a.runsK[3 * r], (size_t)a.runsK[3 * r + 1] * sizeof(double));
        for (int r = 0; r < a.nrunsB; ++r)
          std::memcpy(row + a.runsB[3 * r + 2], a.bd[i] + a.runsB[3 * r], (size_t)a.runsB[3 * r + 1] * sizeof(double));
      }
    }
    a.same_out[i] = same ? 1 : 0;
  }
}
void stage_parallel(const StageArgs& a, int i0, int i1, int nthreads) {
  const int n = i1 - i0;
  const int nt = std::max(1, std::min(std::min(nthreads, 64), n));
  if (nt == 1) { stage_range(a, i0, i1); return; }
  std::vector<std::thread> pool;
  pool.reserve((size_t)nt);
  int started = 0;
  try {                           // (no exception may cross the C ABI: what could not be started runs here)
    for (; started < nt; ++started)
      pool.emplace_back(stage_range, std::cref(a), i0 + (int)((int64_t)n * started / nt), i0 + (int)((int64_t)n * (started + 1) / nt));
  } catch (...) {
  }
  if (started < nt) stage_range(a, i0 + (int)((int64_t)n * started / nt), i1);
  for (auto& th : pool) th.join();
}

bool stage_args_ok(int nblocks, const StageArgs& a, int64_t need) {
  if (nblocks < 0 || !a.same_out) return false;
  if (nblocks > 0 && (!a.kr || !a.kc || !a.kd || !a.knnz || !a.br || !a.bc || !a.bd || !a.bnnz || !a.staging || !a.slots)) return false;
  if ((a.nrunsK > 0 && !a.runsK) || (a.nrunsB > 0 && !a.runsB)) return false;
  for (int r = 0; r < a.nrunsK; ++r)
    if (a.runsK[3 * r] < 0 || a.runsK[3 * r + 1] < 0 || a.runsK[3 * r] + a.runsK[3 * r + 1] > a.ref_knnz || a.runsK[3 * r + 2] < 0 ||
        a.runsK[3 * r + 2] + a.runsK[3 * r + 1] > a.row_stride) return false;
  for (int r = 0; r < a.nrunsB; ++r)
    if (a.runsB[3 * r] < 0 || a.runsB[3 * r + 1] < 0 || a.runsB[3 * r] + a.runsB[3 * r + 1] > a.ref_bnnz || a.runsB[3 * r + 2] < 0 ||
        a.runsB[3 * r + 2] + a.runsB[3 * r + 1] > a.row_stride) return false;
  return need <= a.row_stride;
}
// item(i) for i < nitems on the handle's host threads (items claimed in ascending order), and copy(r0, r1) on the
// caller's thread for every finished slice of items -- rows slots[i0] .. slots[i1 - 1] (slots null: row i = item i).
extern "C++" {
template <class Item, class Copy>
int sliced_upload(pp_handle h, int nitems, int nthreads, const int32_t* slots, Item item, Copy copy) {
  if (nitems <= 0) return 0;
  const int slice = 64, nsl = (nitems + slice - 1) / slice;
  std::vector<std::atomic<int>> done((size_t)nsl);
  for (auto& d : done) d.store(0, std::memory_order_relaxed);
  std::atomic<int> next{0};
  auto work = [&]() {
    for (;;) {
      const int i = next.fetch_add(1, std::memory_order_relaxed);
      if (i >= nitems) break;
      item(i);
      done[(size_t)(i / slice)].fetch_add(1, std::memory_order_release);
    }
  };
  const int nt = std::max(1, std::min(std::min(nthreads, 64), nitems));
  const int started = nt > 1 ? h->stage_pool.start(nt, work) : 0;
  if (started == 0) work();
  hipError_t err = hipSuccess;
  for (int s = 0; s < nsl; ++s) {
    const int i0 = s * slice, i1 = std::min(nitems, i0 + slice);
    while (done[(size_t)s].load(std::memory_order_acquire) < i1 - i0) std::this_thread::yield();
    if (err == hipSuccess) err = copy(slots ? slots[i0] : i0, slots ? slots[i1 - 1] + 1 : i1);
  }
  if (started > 0) h->stage_pool.wait();      // (work refers to this frame)
  if (err != hipSuccess) return fail(h, 3, std::string("host boundary copy: ") + hipGetErrorString(err));
  return 0;
}
}  // extern "C++"
}  // namespace

int pp_stage_values(int nblocks, int nthreads, const int32_t* const* kr, const int32_t* const* kc,
                    const double* const* kd, const int64_t* knnz, const int32_t* const* br, const int32_t* const* bc,
                    const double* const* bd, const int64_t* bnnz, const int32_t* ref_kr, const int32_t* ref_kc,
                    int64_t ref_knnz, const int32_t* ref_br, const int32_t* ref_bc, int64_t ref_bnnz, double* staging,
                    int64_t row_stride, const int32_t* slots, uint8_t* same_out) {
  const StageArgs a{kr, kc, kd, knnz, br, bc, bd, bnnz, ref_kr, ref_kc, ref_knnz, ref_br, ref_bc, ref_bnnz, 0, nullptr, 0,
                    nullptr, staging, row_stride, slots, same_out};
  if (!stage_args_ok(nblocks, a, ref_knnz + ref_bnnz)) return 3;
  stage_parallel(a, 0, nblocks, nthreads);
  return 0;
}

int pp_stage_values_runs(int nblocks, int nthreads, const int32_t* const* kr, const int32_t* const* kc,
                         const double* const* kd, const int64_t* knnz, const int32_t* const* br, const int32_t* const* bc,
                         const double* const* bd, const int64_t* bnnz, const int32_t* ref_kr, const int32_t* ref_kc,
                         int64_t ref_knnz, const int32_t* ref_br, const int32_t* ref_bc, int64_t ref_bnnz, int nruns_k,
                         const int64_t* runs_k, int nruns_b, const int64_t* runs_b, double* staging, int64_t row_stride,
                         const int32_t* slots, uint8_t* same_out) {
  const StageArgs a{kr, kc, kd, knnz, br, bc, bd, bnnz, ref_kr, ref_kc, ref_knnz, ref_br, ref_bc, ref_bnnz, nruns_k, runs_k,
                    nruns_b, runs_b, staging, row_stride, slots, same_out};
  if (!runs_k || !stage_args_ok(nblocks, a, 0)) return 3;
  stage_parallel(a, 0, nblocks, nthreads);
  return 0;
}

// The same with the upload overlapped: the blocks (ascending slots) are staged in slices and every finished slice of
// rows goes to the device with an asynchronous copy while the host threads stage the next one (the staging array
// must be pinned for the copies to be asynchronous).  The rows of blocks reported in same_out as not staged are
// uploaded by the caller afterwards (pp_upload_values_compact on their row range).
int pp_stage_upload_compact(pp_handle h, int group, int nblocks, int nthreads, const int32_t* const* kr,
                            const int32_t* const* kc, const double* const* kd, const int64_t* knnz,
                            const int32_t* const* br, const int32_t* const* bc, const double* const* bd,
                            const int64_t* bnnz, const int32_t* ref_kr, const int32_t* ref_kc, int64_t ref_knnz,
                            const int32_t* ref_br, const int32_t* ref_bc, int64_t ref_bnnz, int nruns_k, const int64_t* runs_k,
                            int nruns_b, const int64_t* runs_b, double* staging, const int32_t* slots, uint8_t* same_out) {
  Group* g = get_group(h, group);
  if (h) { if (int rc = stage_job_finish(h)) return rc; }      // (the host threads serve one job at a time)
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_stage_upload_compact: bad group or symbolic phase not finished");
  const StageArgs a{kr, kc, kd, knnz, br, bc, bd, bnnz, ref_kr, ref_kc, ref_knnz, ref_br, ref_bc, ref_bnnz, nruns_k, runs_k,
                    nruns_b, runs_b, staging, (int64_t)g->nraw_used, slots, same_out};
  if (!runs_k || !stage_args_ok(nblocks, a, 0)) return fail(h, 3, "pp_stage_upload_compact: bad arguments");
  for (int i = 0; i < nblocks; ++i)
    if (slots[i] < 0 || slots[i] >= g->batch || (i > 0 && slots[i] <= slots[i - 1]))
      return fail(h, 3, "pp_stage_upload_compact: slots must be ascending and inside the batch");
  PP_HIP(hipSetDevice(h->device));
  if (int rc = ensure_optional(h, g, OPT_RAW)) return rc;
  g->input_mode = Group::IN_COMPACT;
  const size_t stride = (size_t)g->nraw_used;
  if (g->staged_row_valid.size() != (size_t)g->batch || g->stage_host != staging) g->staged_row_valid.assign((size_t)g->batch, 0);
  g->stage_host = staging;
  // (whole slices are sent: every row in a slice's range mirrors the staging array afterwards -- also the rows of blocks
  // that were NOT staged, whose stale staging rows the caller overwrites and sends again)
  for (int i = 0; i < nblocks; ++i) g->staged_row_valid[(size_t)slots[i]] = 0;
  const int rc = sliced_upload(h, nblocks, nthreads, slots, [&](int i) { stage_range(a, i, i + 1); },
                               [&](int r0, int r1) {
                                 return stride == 0 ? hipSuccess
                                                    : hipMemcpyAsync(g->raw_own + (size_t)r0 * stride, staging + (size_t)r0 * stride,
                                                                     (size_t)(r1 - r0) * stride * sizeof(double), hipMemcpyHostToDevice, h->stream);
                               });
  if (rc == 0)
    for (int i = 0; i < nblocks; ++i) g->staged_row_valid[(size_t)slots[i]] = same_out[i];
  return rc;
}

// The same for blocks whose index arrays were verified at an earlier call (nothing is compared), WITHOUT waiting: the
// arguments are copied, the host threads start staging and send every slice they finish themselves, and the call
// returns -- the caller prepares its next batch of blocks meanwhile.  A second begin first waits for the job in flight;
// pp_stage_upload_end waits for the last one and reports the first error.  Nothing else may be enqueued in between.
namespace {
struct StageJob {
  std::vector<const double*> kd, bd;
  std::vector<int32_t> slots;
  std::vector<int64_t> runs_k, runs_b;
  std::vector<std::atomic<int>> done;
  std::atomic<int> next{0};
  std::atomic<int> err{0};
  int nblocks = 0, slice = 64;
  double *staging = nullptr, *dev = nullptr;
  size_t stride = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  // Compare while staging: a row whose staged values mirror the device (valid) is compared piece by piece (CH doubles of
  // the compact row) with the new values; only pieces that differ are copied into the staging row and marked, and a
  // slice sends the column ranges some row of it marked -- constant Jacobian values (80 % of a C3 block) do not cross
  // PCIe again.  Rows that do not mirror the device yet are copied and sent whole.
  static constexpr size_t CH = 512;
  uint8_t* row_valid = nullptr;          // the group's staged_row_valid
  size_t nchunk = 0;
  std::vector<uint8_t> changed;          // [block][chunk]
  std::atomic<long long> sent_bytes{0};
  bool compare = true;
  void stage_run(double* row, const double* src, size_t d, size_t len, bool valid, uint8_t* mark) {
    size_t p0 = d;
    const size_t end = d + len;
    while (p0 < end) {
      const size_t p1 = std::min(end, (p0 / CH + 1) * CH);
      const size_t bytes = (p1 - p0) * sizeof(double);
      if (!valid || std::memcmp(row + p0, src + (p0 - d), bytes) != 0) {
        std::memcpy(row + p0, src + (p0 - d), bytes);
        mark[p0 / CH] = 1;
      }
      p0 = p1;
    }
  }
  void work() {
    bool device_set = false;
    for (;;) {
      const int i = next.fetch_add(1, std::memory_order_relaxed);
      if (i >= nblocks) break;
      const size_t slot = (size_t)slots[(size_t)i];
      double* row = staging + slot * stride;
      const bool valid = compare && row_valid[slot] != 0;
      uint8_t* mark = changed.data() + (size_t)i * nchunk;
      for (size_t r = 0; r + 2 < runs_k.size() + 1; r += 3) stage_run(row, kd[(size_t)i] + runs_k[r], (size_t)runs_k[r + 2], (size_t)runs_k[r + 1], valid, mark);
      for (size_t r = 0; r + 2 < runs_b.size() + 1; r += 3) stage_run(row, bd[(size_t)i] + runs_b[r], (size_t)runs_b[r + 2], (size_t)runs_b[r + 1], valid, mark);
      const int s = i / slice, i0 = s * slice, i1 = std::min(nblocks, i0 + slice);
      if (done[(size_t)s].fetch_add(1, std::memory_order_acq_rel) + 1 == i1 - i0 && stride > 0) {
        // the last block of the slice: its rows go to the device (rows of blocks outside the call lie in between only
        // when the caller mixes paths; they are sent again by whoever stages them)
        if (!device_set) { (void)hipSetDevice(device); device_set = true; }
        const size_t r0 = (size_t)slots[(size_t)i0], r1 = (size_t)slots[(size_t)i1 - 1] + 1;
        bool all_valid = compare;
        for (int q = i0; q < i1 && all_valid; ++q) all_valid = row_valid[(size_t)slots[(size_t)q]] != 0;
        hipError_t e = hipSuccess;
        if (!all_valid) {
          e = hipMemcpyAsync(dev + r0 * stride, staging + r0 * stride, (r1 - r0) * stride * sizeof(double), hipMemcpyHostToDevice, stream);
          sent_bytes.fetch_add((long long)((r1 - r0) * stride * sizeof(double)), std::memory_order_relaxed);
        } else {
          // union of the marked pieces over the rows of the slice -> column ranges -> one 2-D copy per range
          std::vector<uint8_t> u(nchunk, 0);
          for (int q = i0; q < i1; ++q) {
            const uint8_t* m = changed.data() + (size_t)q * nchunk;
            for (size_t c = 0; c < nchunk; ++c) u[c] |= m[c];
          }
          for (size_t c = 0; c < nchunk && e == hipSuccess;) {
            if (!u[c]) { ++c; continue; }
            size_t c1 = c;
            while (c1 < nchunk && u[c1]) ++c1;
            const size_t col0 = c * CH, col1 = std::min(stride, c1 * CH);
            e = hipMemcpy2DAsync(dev + r0 * stride + col0, stride * sizeof(double), staging + r0 * stride + col0, stride * sizeof(double),
                                 (col1 - col0) * sizeof(double), r1 - r0, hipMemcpyHostToDevice, stream);
            sent_bytes.fetch_add((long long)((col1 - col0) * sizeof(double) * (r1 - r0)), std::memory_order_relaxed);
            c = c1;
          }
        }
        if (e != hipSuccess) { int zero = 0; err.compare_exchange_strong(zero, (int)e); }
        else for (int q = i0; q < i1; ++q) row_valid[(size_t)slots[(size_t)q]] = 1;
      }
    }
  }
};
int stage_job_finish(pp_handle h) {
  StageJob* j = (StageJob*)h->stage_job;
  if (!j) return 0;
  h->stage_pool.wait();
  const int e = j->err.load();
  delete j;
  h->stage_job = nullptr;
  if (e != 0) return fail(h, 3, std::string("pp_stage_upload_verified_begin: copy failed: ") + hipGetErrorString((hipError_t)e));
  return 0;
}
}  // namespace

int pp_stage_upload_verified_begin(pp_handle h, int group, int nblocks, int nthreads, const double* const* kd,
                                   const double* const* bd, int64_t ref_knnz, int64_t ref_bnnz, int nruns_k, const int64_t* runs_k,
                                   int nruns_b, const int64_t* runs_b, double* staging, const int32_t* slots) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_stage_upload_verified_begin: bad group or symbolic phase not finished");
  if (int rc = stage_job_finish(h)) return rc;
  if (nblocks <= 0) return 0;
  StageArgs a{};
  a.ref_knnz = ref_knnz; a.ref_bnnz = ref_bnnz;
  a.nrunsK = nruns_k; a.runsK = runs_k; a.nrunsB = nruns_b; a.runsB = runs_b; a.row_stride = (int64_t)g->nraw_used;
  uint8_t dummy = 0;
  a.same_out = &dummy;
  if (!kd || !bd || !staging || !slots || !runs_k || ref_knnz < 0 || ref_bnnz < 0 || ref_knnz + ref_bnnz != g->nraw || !stage_args_ok(0, a, 0)) return fail(h, 3, "pp_stage_upload_verified_begin: bad arguments");
  for (int i = 0; i < nblocks; ++i)
    if (slots[i] < 0 || slots[i] >= g->batch || (i > 0 && slots[i] <= slots[i - 1]) || !kd[i] || (a.ref_bnnz > 0 && nruns_b > 0 && !bd[i]))
      return fail(h, 3, "pp_stage_upload_verified_begin: slots must be ascending and inside the batch, data pointers non-null");
  PP_HIP(hipSetDevice(h->device));
  if (int rc = ensure_optional(h, g, OPT_RAW)) return rc;
  g->input_mode = Group::IN_COMPACT;
  StageJob* j = new (std::nothrow) StageJob;
  if (!j) return fail(h, 3, "pp_stage_upload_verified_begin: out of host memory");
  try {
    j->kd.assign(kd, kd + nblocks); j->bd.assign(bd, bd + nblocks); j->slots.assign(slots, slots + nblocks);
    j->runs_k.assign(runs_k, runs_k + 3 * (size_t)nruns_k);
    if (nruns_b > 0) j->runs_b.assign(runs_b, runs_b + 3 * (size_t)nruns_b);
    j->done = std::vector<std::atomic<int>>((size_t)((nblocks + j->slice - 1) / j->slice));
  } catch (...) { delete j; return fail(h, 3, "pp_stage_upload_verified_begin: out of host memory"); }
  for (auto& d : j->done) d.store(0, std::memory_order_relaxed);
  j->nblocks = nblocks; j->staging = staging; j->dev = g->raw_own; j->stride = (size_t)g->nraw_used;
  j->device = h->device; j->stream = h->stream;
  if (g->staged_row_valid.size() != (size_t)g->batch || g->stage_host != staging) g->staged_row_valid.assign((size_t)g->batch, 0);
  g->stage_host = staging;
  j->row_valid = g->staged_row_valid.data();
  j->nchunk = (j->stride + StageJob::CH - 1) / StageJob::CH;
  static const bool no_compare = pp::env_switch("PP_NO_STAGE_COMPARE") != nullptr;      // (measurement switch)
  j->compare = !no_compare;
  try { j->changed.assign((size_t)nblocks * j->nchunk, 0); } catch (...) { delete j; return fail(h, 3, "pp_stage_upload_verified_begin: out of host memory"); }
  h->stage_job = j;
  const int nt = std::max(1, std::min(std::min(nthreads, 64), nblocks));
  if (h->stage_pool.start(nt, [j]() { j->work(); }) == 0) {      // no thread could be started: here and now
    j->work();
    const int e = j->err.load();
    delete j;
    h->stage_job = nullptr;
    if (e != 0) return fail(h, 3, std::string("pp_stage_upload_verified_begin: copy failed: ") + hipGetErrorString((hipError_t)e));
  }
  return 0;
}

int pp_stage_upload_end(pp_handle h) { return h ? stage_job_finish(h) : 3; }

// The right-hand sides of ALL blocks of a group (src[i]: the n values of the block in slot i) through the pinned staging
// array [batch][n] to the device, the copy of a slice of rows overlapping the host threads' work on the next one.
int pp_upload_rhs_rows(pp_handle h, int group, int nrows, int nthreads, const double* const* src, double* staging) {
  Group* g = get_group(h, group);
  if (h) { if (int rc = stage_job_finish(h)) return rc; }      // (the host threads serve one job at a time)
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_upload_rhs_rows: bad group or symbolic phase not finished");
  if (nrows != g->batch || !src || !staging) return fail(h, 3, "pp_upload_rhs_rows: one source row per block of the group is needed");
  PP_HIP(hipSetDevice(h->device));
  if (int rc = alloc_value_storage(h)) return rc;
  if (!g->dev.rhs) { if (int rc = ensure_optional(h, g, OPT_RHS)) return rc; }
  const size_t n = (size_t)g->plan.n;
  return sliced_upload(h, nrows, nthreads, nullptr, [&](int i) { std::memcpy(staging + (size_t)i * n, src[i], n * sizeof(double)); },
                       [&](int r0, int r1) {
                         return hipMemcpyAsync(g->dev.rhs + (size_t)r0 * n, staging + (size_t)r0 * n, (size_t)(r1 - r0) * n * sizeof(double),
                                               hipMemcpyHostToDevice, h->stream);
                       });
}

// The solutions of a group ([batch][n]) to the host.  dst null: one asynchronous copy into the pinned array (the caller
// hands out its rows and synchronises, pp_synchronize, before they are read).  dst given (pageable memory, e.g. a fresh
// array per call): the copy goes slice by slice through the pinned array and host threads move every slice that has
// arrived on to dst while the next one is in flight (they also take the page faults of a fresh dst, side by side);
// returns when dst is complete.
int pp_download_solution_rows(pp_handle h, int group, int nthreads, double* pinned, double* dst) {
  Group* g = get_group(h, group);
  if (h) { if (int rc = stage_job_finish(h)) return rc; }      // (the host threads serve one job at a time)
  if (!g || !h->symbolic_done || !pinned) return fail(h, 3, "pp_download_solution_rows: bad group or no pinned array");
  if (!g->dev.xout) return fail(h, 3, "pp_download_solution_rows: no solution in the [instance][row] layout (native vectors bound?)");
  PP_HIP(hipSetDevice(h->device));
  const size_t n = (size_t)g->plan.n;
  const int nrows = g->batch;
  if (!dst) {
    PP_HIP(hipMemcpyAsync(pinned, g->dev.xout, (size_t)nrows * n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    return 0;
  }
  const int slice = 64, nsl = (nrows + slice - 1) / slice;
  while ((int)h->dl_events.size() < nsl) {
    hipEvent_t e = nullptr;
    PP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    h->dl_events.push_back(e);
  }
  for (int s = 0; s < nsl; ++s) {
    const size_t r0 = (size_t)s * slice, r1 = std::min((size_t)nrows, r0 + slice);
    PP_HIP(hipMemcpyAsync(pinned + r0 * n, g->dev.xout + r0 * n, (r1 - r0) * n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    PP_HIP(hipEventRecord(h->dl_events[(size_t)s], h->stream));
  }
  std::atomic<int> ready{0}, next{0};
  std::atomic<bool> failed{false};
  auto work = [&]() {
    for (;;) {
      const int i = next.fetch_add(1, std::memory_order_relaxed);
      if (i >= nrows) break;
      while (ready.load(std::memory_order_acquire) <= i / slice && !failed.load(std::memory_order_relaxed)) std::this_thread::yield();
      if (failed.load(std::memory_order_relaxed)) continue;
      std::memcpy(dst + (size_t)i * n, pinned + (size_t)i * n, n * sizeof(double));
    }
  };
  const int nt = std::max(1, std::min(std::min(nthreads, 64), nrows));
  const int started = nt > 1 ? h->stage_pool.start(nt, work) : 0;
  hipError_t err = hipSuccess;
  for (int s = 0; s < nsl && err == hipSuccess; ++s) {
    err = hipEventSynchronize(h->dl_events[(size_t)s]);
    if (err == hipSuccess) ready.store(s + 1, std::memory_order_release);
  }
  if (err != hipSuccess) failed.store(true);
  if (started > 0) h->stage_pool.wait(); else work();
  if (err != hipSuccess) return fail(h, 3, std::string("pp_download_solution_rows: ") + hipGetErrorString(err));
  return 0;
}

// dst[idx[i]][0 .. row_doubles) = src[i][0 .. row_doubles): the right-hand sides of the local blocks into their staging
// rows, on host threads (75 MB per back-solve at the headline size)
int pp_copy_rows(int nrows, int nthreads, const double* const* src, const int64_t* idx, double* dst, int64_t row_doubles) {
  if (nrows < 0 || row_doubles < 0 || (nrows > 0 && (!src || !idx || !dst))) return 3;
  auto work = [&](int i0, int i1) {
    for (int i = i0; i < i1; ++i) std::memcpy(dst + (size_t)idx[i] * (size_t)row_doubles, src[i], (size_t)row_doubles * sizeof(double));
  };
  const int nt = std::max(1, std::min(std::min(nthreads, 64), nrows));
  if (nt == 1) { work(0, nrows); return 0; }
  std::vector<std::thread> pool;
  pool.reserve((size_t)nt);
  int started = 0;
  try {
    for (; started < nt; ++started)
      pool.emplace_back(work, (int)((int64_t)nrows * started / nt), (int)((int64_t)nrows * (started + 1) / nt));
  } catch (...) {
  }
  if (started < nt) work((int)((int64_t)nrows * started / nt), nrows);
  for (auto& th : pool) th.join();
  return 0;
}

// pinned host memory for staging arrays / result buffers of the host boundary (hipHostMalloc; NULL on failure)
// (bounded: page-locked memory is taken from what the whole machine can use -- a caller that keeps asking, e.g. many
// solver objects with result pools, gets NULL beyond PP_PINNED_LIMIT_MB (default 16 GiB per process) and falls back to
// pageable arrays, instead of driving the host out of lockable memory)
namespace {
std::mutex g_pin_mu;
std::map<void*, int64_t> g_pin_sizes;
int64_t g_pin_total = 0;
int64_t pin_limit() {
  static const int64_t lim = [] {
    const char* e = pp::env_switch("PP_PINNED_LIMIT_MB");
    const int64_t mb = e ? std::atoll(e) : 16384;
    return (mb > 0 ? mb : 16384) * (int64_t)1048576;
  }();
  return lim;
}
}  // namespace

void* pp_host_alloc(int64_t bytes) {
  void* p = nullptr;
  if (bytes <= 0) return nullptr;
  {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    if (g_pin_total + bytes > pin_limit()) return nullptr;
    g_pin_total += bytes;
  }
  if (hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pin_total -= bytes;
    return nullptr;
  }
  std::lock_guard<std::mutex> lk(g_pin_mu);
  g_pin_sizes[p] = bytes;
  return p;
}

void pp_host_free(void* p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    auto it = g_pin_sizes.find(p);
    if (it != g_pin_sizes.end()) { g_pin_total -= it->second; g_pin_sizes.erase(it); }
  }
  (void)hipHostFree(p);
}

int pp_set_pivot_tolerance(pp_handle h, double u_symbolic, double u_runtime) {
  if (!h) return 3;
  if (u_symbolic < 0.0 || u_symbolic > 0.5 || u_runtime < 0.0 || u_runtime > 0.5)
    return fail(h, 3, "pivot tolerances must lie in [0, 0.5] (0: default / off)");
  h->pivot_threshold = u_symbolic;
  h->growth_bound = u_runtime > 0.0 ? 1.0 / u_runtime : 1e8;
  h->growth_fatal = u_runtime > 0.0;
  return 0;
}

int pp_get_growth_count(pp_handle h, int64_t* out) {
  if (!h || !h->schur_done || !out) return fail(h, 3, "pp_get_growth_count before pp_factor_schur");
  *out = (int64_t)h->status_host[5];     // (valid once pp_get_status has seen the mailbox of this factorisation)
  return 0;
}

int pp_find_growth(pp_handle h, int group, int32_t* instance_out) {
  Group* g = get_group(h, group);
  if (!g || !instance_out || !h->numeric_done) return fail(h, 3, "pp_find_growth: bad group or no numeric factorization");
  const GroupDev& d = g->dev;
  *instance_out = -1;
  PP_HIP(hipSetDevice(h->device));
  PP_HIP(hipStreamSynchronize(h->stream));
  std::vector<int> flags((size_t)d.bpad);
  PP_HIP(hipMemcpy(flags.data(), d.growth + d.bpad, flags.size() * sizeof(int), hipMemcpyDeviceToHost));
  for (int b = 0; b < d.batch; ++b)
    if (flags[(size_t)b]) { *instance_out = b; break; }
  return 0;
}

int pp_find_zero_pivot(pp_handle h, int group, int32_t* instance_out) {
  Group* g = get_group(h, group);
  if (!g || !instance_out || !h->numeric_done) return fail(h, 3, "pp_find_zero_pivot: bad group or no numeric factorization");
  const GroupDev& d = g->dev;
  *instance_out = -1;
  PP_HIP(hipSetDevice(h->device));
  PP_HIP(hipStreamSynchronize(h->stream));
  // rare path (a factorisation that reported numerically zero pivots): the 16-bit codes come to the host as they are
  std::vector<unsigned short> codes((size_t)g->plan.npiv * d.bpad);
  PP_HIP(hipMemcpy(codes.data(), d.codes, codes.size() * sizeof(unsigned short), hipMemcpyDeviceToHost));
  for (int p = 0; p < g->plan.npiv && *instance_out < 0; ++p)        // first pivot in elimination order that broke
    for (int b = 0; b < d.batch; ++b)
      if ((codes[(size_t)p * d.bpad + b] >> 8) & 15u) { *instance_out = b; break; }
  return 0;
}

int pp_get_factor(pp_handle h, int group, int which, int instance, double* out, int64_t count) {
  Group* g = get_group(h, group);
  if (!g) return fail(h, 3, "pp_get_factor: bad group");
  const GroupDev& d = g->dev;
  const double* src = which == 0 ? d.U : which == 1 ? d.L : which == 2 ? d.Dinv : which == 3 ? d.rawT : nullptr;
  if (!src) return fail(h, 3, "pp_get_factor: that array does not exist (fused sources: no transposed input)");
  const int64_t rows = which == 2 ? g->plan.dsize : which == 3 ? g->nraw_used : g->plan.usize;
  if (!src || instance < 0 || instance >= d.batch || count > rows) return fail(h, 3, "pp_get_factor: bad arguments");
  PP_HIP(hipSetDevice(h->device));
  PP_HIP(hipStreamSynchronize(h->stream));
  PP_HIP(hipMemcpy2D(out, sizeof(double), src + instance, sizeof(double) * (size_t)d.bpad, sizeof(double), (size_t)count,
                     hipMemcpyDeviceToHost));
  return 0;
}

// ---- RCCL, opened at run time (no link-time dependency): the two data-path all-reduces on the handle's stream
namespace {
typedef int (*fn_get_id)(void*);
typedef int (*fn_init_rank)(void**, int, ppd_nccl_id, int);
typedef int (*fn_destroy)(void*);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_allgather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*fn_errstr)(int);
struct Rccl {
  void* lib = nullptr;
  fn_get_id get_id = nullptr; fn_init_rank init_rank = nullptr; fn_destroy destroy = nullptr;
  fn_allreduce allreduce = nullptr; fn_allgather allgather = nullptr; fn_errstr errstr = nullptr;
  bool tried = false;
};
Rccl g_rccl;
std::mutex g_rccl_mu;
bool rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.tried) return g_rccl.allreduce != nullptr;
  g_rccl.tried = true;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.lib) break;
  }
  if (!g_rccl.lib) return false;
  g_rccl.get_id = (fn_get_id)dlsym(g_rccl.lib, "ncclGetUniqueId");
  g_rccl.init_rank = (fn_init_rank)dlsym(g_rccl.lib, "ncclCommInitRank");
  g_rccl.destroy = (fn_destroy)dlsym(g_rccl.lib, "ncclCommDestroy");
  g_rccl.allreduce = (fn_allreduce)dlsym(g_rccl.lib, "ncclAllReduce");
  g_rccl.allgather = (fn_allgather)dlsym(g_rccl.lib, "ncclAllGather");
  g_rccl.errstr = (fn_errstr)dlsym(g_rccl.lib, "ncclGetErrorString");
  if (!g_rccl.get_id || !g_rccl.init_rank || !g_rccl.destroy || !g_rccl.allreduce) { g_rccl.allreduce = nullptr; return false; }
  return true;
}
void rccl_release(pp_handle h) {
  if (h->rccl_comm && g_rccl.destroy) (void)g_rccl.destroy(h->rccl_comm);
  h->rccl_comm = nullptr;
  h->rccl_ranks = 0;
}
extern "C++" std::string rccl_msg(const char* what, int rc) {
  return std::string(what) + ": " + (g_rccl.errstr ? g_rccl.errstr(rc) : "RCCL error") + " (" + std::to_string(rc) + ")";
}
}  // namespace

int pp_comm_unique_id(uint8_t id_out[128]) {
  if (!id_out || !rccl_load()) return 3;
  ppd_nccl_id id;
  if (g_rccl.get_id(&id) != 0) return 3;
  std::memcpy(id_out, id.internal, 128);
  return 0;
}

int pp_comm_init(pp_handle h, int nranks, int rank, const uint8_t id[128]) {
  if (!h || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(h, 3, "pp_comm_init: bad arguments");
  if (!rccl_load()) return fail(h, 3, "pp_comm_init: librccl could not be opened");
  PP_HIP(hipSetDevice(h->device));
  if (h->rccl_comm) { (void)g_rccl.destroy(h->rccl_comm); h->rccl_comm = nullptr; h->rccl_ranks = 0; }
  ppd_nccl_id uid;
  std::memcpy(uid.internal, id, 128);
  void* comm = nullptr;
  const int rc = g_rccl.init_rank(&comm, nranks, uid, rank);
  if (rc != 0) return fail(h, 3, rccl_msg("ncclCommInitRank", rc));
  h->rccl_comm = comm;
  h->rccl_ranks = nranks;
  h->rccl_rank = rank;
  return 0;
}

int pp_comm_size(pp_handle h) { return h ? h->rccl_ranks : 0; }

int pp_allreduce_schur(pp_handle h) {
  if (!h || !h->rccl_comm) return fail(h, 3, "pp_allreduce_schur: no communicator (pp_comm_init)");
  if (!h->numeric_done) return fail(h, 3, "pp_allreduce_schur before pp_numeric_local");
  PP_HIP(hipSetDevice(h->device));
  const size_t count = schur_doubles(h) + PP_TAIL;
  // (behind the Schur update, wherever it was enqueued)
  const int rc = g_rccl.allreduce(h->S, h->S, count, /* ncclDouble */ 8, /* ncclSum */ 0, h->rccl_comm,
                                  h->schur_on_side ? h->dense_stream : h->stream);
  if (rc != 0) return fail(h, 3, rccl_msg("ncclAllReduce(S)", rc));
  if (h->schur_on_side) {
    // every other collective of this communicator goes to the handle's stream: order them behind this one (two
    // collectives of one communicator in flight on two streams may be executed in different orders by different ranks)
    if (!h->ev_coll_side) PP_HIP(hipEventCreateWithFlags(&h->ev_coll_side, hipEventDisableTiming));
    PP_HIP(hipEventRecord(h->ev_coll_side, h->dense_stream));
    h->coll_side_pending = true;
  }
  return 0;
}

// collectives on the handle's stream wait for a collective that was enqueued on the side stream
static int order_behind_side_collective(pp_handle h) {
  if (h->coll_side_pending) {
    PP_HIP(hipStreamWaitEvent(h->stream, h->ev_coll_side, 0));
    h->coll_side_pending = false;
  }
  return 0;
}

int pp_allreduce_rs(pp_handle h) {
  if (!h || !h->rccl_comm) return fail(h, 3, "pp_allreduce_rs: no communicator (pp_comm_init)");
  PP_HIP(hipSetDevice(h->device));
  if (h->nc == 0) return 0;
  if (int rc = order_behind_side_collective(h)) return rc;
  const int rc = g_rccl.allreduce(h->rs, h->rs, (size_t)h->nc, 8, 0, h->rccl_comm, h->stream);
  if (rc != 0) return fail(h, 3, rccl_msg("ncclAllReduce(r_s)", rc));
  return 0;
}

// (refine.hip: the sums of the coupling rows and one slot per rank of the a-posteriori check, in place, on the handle's stream)
int ppi_allreduce_sum(pp_handle h, double* buf, size_t count) {
  if (!h || !h->rccl_comm) return fail(h, 3, "ppi_allreduce_sum: no communicator (pp_comm_init)");
  if (count == 0) return 0;
  if (int rc = order_behind_side_collective(h)) return rc;
  const int rc = g_rccl.allreduce(buf, buf, count, 8, 0, h->rccl_comm, h->stream);
  if (rc != 0) return fail(h, 3, rccl_msg("ncclAllReduce(check)", rc));
  return 0;
}

int pp_comm_allgather(pp_handle h, const double* src, double* table, int64_t count) {
  if (!h || !h->rccl_comm || !g_rccl.allgather) return fail(h, 3, "pp_comm_allgather: no communicator (pp_comm_init)");
  if (!src || !table || count < 0) return fail(h, 3, "pp_comm_allgather: bad arguments");
  PP_HIP(hipSetDevice(h->device));
  if (count == 0) return 0;
  if (int rc = order_behind_side_collective(h)) return rc;
  const int rc = g_rccl.allgather(src, table, (size_t)count, /* ncclDouble */ 8, h->rccl_comm, h->stream);
  if (rc != 0) return fail(h, 3, rccl_msg("ncclAllGather", rc));
  return 0;
}

}  // extern "C"
