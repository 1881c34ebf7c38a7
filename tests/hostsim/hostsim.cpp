// TEST-ONLY host interpreter of the symbolic plan (never shipped, never a fallback).
// Executes exactly the task records the HIP kernels execute (same order, same
// formulas, one instance at a time) so the schedule can be validated without a GPU.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>

#include "../../parapint_amd/csrc/plan.hpp"
#include "../../parapint_amd/csrc/dense_bk.hpp"

using pp::Plan;

namespace {
struct HostCtx {
  int tid() const { return 0; }
  int nthreads() const { return 1; }
  void sync() {}
  void argmax(double v, int i, double* vmax, int* imax) { *vmax = v; *imax = i; }
  double maxval(double v) { return v; }
  double sum(double v) { return v; }
};
}  // namespace

static int g_sn_wmax = 0, g_sn_tol = -1, g_batch_hint = 0, g_mapped_hint = 0;
static double g_growth_bound = 1e8, g_pivot_threshold = 0.0;   // as pp_set_pivot_tolerance
static int g_last_growth = 0, g_growth_fatal = 0;
static int g_first_zero_piv = -1;      // first block pivot of the last ppsim_factor call that held a numerically zero sub-pivot (diagnostic)

extern "C" {

// test knob: supernode width cap / padded-row tolerance for plans created afterwards (0 / -1: defaults)
void ppsim_set_supernodes(int wmax, int tol) { g_sn_wmax = wmax; g_sn_tol = tol; }
// instances of the pattern group the next plans are made for (0: unknown = the large-batch task sizes), plan.hpp:tune_for_batch
void ppsim_set_batch_hint(int batch) { g_batch_hint = batch; }
// the next plans are made for a group with coupling rows of its own per instance (plan.hpp:tune_for_mapped_group)
void ppsim_set_mapped_hint(int mapped) { g_mapped_hint = mapped; }
void ppsim_set_pivot_tolerance(double u_symbolic, double u_runtime) {
  g_pivot_threshold = u_symbolic;
  g_growth_bound = u_runtime > 0.0 ? 1.0 / u_runtime : 1e8;
  g_growth_fatal = u_runtime > 0.0;
}
int ppsim_growth_fatal() { return g_growth_fatal; }
int ppsim_first_zero_pivot() { return g_first_zero_piv; }
int ppsim_last_growth() { return g_last_growth; }

void* ppsim_create(int n, int nc, int nnzK, const int* rowK, const int* colK, int nnzB, const int* rowB,
                   const int* colB, const double* vals, int max_entries, int delta_abs, double delta_rel) {
  auto* P = new Plan();
  pp::PlanOptions opt;
  pp::tune_for_batch(opt, g_batch_hint);
  if (g_mapped_hint) pp::tune_for_mapped_group(opt, g_batch_hint);
  if (max_entries > 0) opt.max_task_entries = max_entries;
  if (delta_abs >= 0) opt.md_delta_abs = delta_abs;
  if (delta_rel >= 0) opt.md_delta_rel = delta_rel;
  if (g_sn_wmax > 0) opt.sn_wmax = g_sn_wmax;
  if (g_sn_tol >= 0) opt.sn_tol_rows = g_sn_tol;
  if (g_pivot_threshold > 0.0) opt.pivot_threshold = g_pivot_threshold;
  { std::string bad; pp::apply_plan_tune(opt, std::getenv("PP_PLAN_TUNE"), bad); }
  int rc = pp::build_plan(n, nc, nnzK, rowK, colK, nnzB, rowB, colB, vals, opt, *P);
  if (rc != 0) { /* keep the plan so the error string can be read */ }
  return P;
}
void ppsim_destroy(void* h) { delete (Plan*)h; }
const char* ppsim_error(void* h) { return ((Plan*)h)->error.c_str(); }

// out: n, nc, npiv, n_levels, n_2x2, usize, nnz_L, flops_factor, flops_schur, ntasks, nruns, ntiles, ntilerecs
void ppsim_stats(void* h, int64_t* out) {
  Plan& P = *(Plan*)h;
  out[0] = P.n; out[1] = P.nc; out[2] = P.npiv; out[3] = P.n_levels; out[4] = P.n_2x2; out[5] = P.usize;
  out[6] = P.nnz_L; out[7] = P.flops_factor; out[8] = P.flops_schur; out[9] = (int64_t)P.ftasks.size();
  out[4] = P.n_2x2 + 1000000LL * P.tail_level0;
  out[10] = (int64_t)P.fentries.size(); out[11] = (int64_t)P.stile_a.size(); out[12] = (int64_t)P.stile_rec.size();
}
int ppsim_nrowidx(void* h) { return (int)((Plan*)h)->rowidx.size(); }
void ppsim_get_struct(void* h, int* piv_start, int* piv_w, int* rowptr, int* rowidx) {
  Plan& P = *(Plan*)h;
  std::memcpy(piv_start, P.piv_start.data(), sizeof(int) * (P.npiv + 1));
  std::memcpy(piv_w, P.piv_w.data(), sizeof(int) * P.npiv);
  std::memcpy(rowptr, P.piv_rowptr.data(), sizeof(int) * (P.npiv + 1));
  std::memcpy(rowidx, P.rowidx.data(), sizeof(int) * P.rowidx.size());
}
int ppsim_dsize(void* h) { return ((Plan*)h)->dsize; }
// chain fronts: out[0] = fronts, out[1] = panels in fronts, out[2] = factor levels, out[3] = widest front, out[4] = most rows
void ppsim_chain_stats(void* h, int* out) {
  Plan& P = *(Plan*)h;
  out[0] = (int)P.chain_m.size(); out[1] = (int)P.chain_piv.size(); out[2] = P.n_flevels; out[3] = 0; out[4] = 0;
  for (size_t c = 0; c < P.chain_m.size(); ++c) { out[3] = std::max(out[3], P.chain_w[c]); out[4] = std::max(out[4], P.chain_m[c]); }
}
// tile tasks: out[0] = tasks (no padding), out[1] = records, out[2] = operand rows present in the records (of 8 per record),
// out[3] = product entries of the tile tasks' rows, out[4] = multiply-adds of those entries, out[5] = multiply-adds of the records
void ppsim_tile_stats(void* h, long long* out) {
  Plan& P = *(Plan*)h;
  for (int i = 0; i < 6; ++i) out[i] = 0;
  for (auto& t : P.ttasks) {
    if (t.kind < 0) continue;
    out[0]++;
    for (int r = t.te0; r < t.te1; ++r) {
      const int* rec = &P.trec[(size_t)r * PP_TREC_INTS];
      int pres = 0;
      for (int q = 1; q <= 8; ++q) pres += rec[q] >= 0;
      out[1]++; out[2] += pres; out[5] += 16LL * rec[0];
    }
    for (int rr = 0; rr < t.r1 - t.r0; ++rr)
      for (int e = P.fdst_ptr[t.dptr0 + rr]; e < P.fdst_ptr[t.dptr0 + rr + 1]; ++e)
        if (P.fentries[e].u >= 0) { out[3]++; out[4] += (P.fentries[e].q >> 4) & 15; }
  }
}
// per front: level, panels, m, W  (4 ints each)
void ppsim_chain_fronts(void* h, int* out) {
  Plan& P = *(Plan*)h;
  for (size_t c = 0; c < P.chain_m.size(); ++c) {
    out[4 * c] = P.chain_level[c]; out[4 * c + 1] = P.chain_ptr[c + 1] - P.chain_ptr[c]; out[4 * c + 2] = P.chain_m[c]; out[4 * c + 3] = P.chain_w[c];
  }
}
void ppsim_get_perm(void* h, int* perm) { Plan& P = *(Plan*)h; std::memcpy(perm, P.perm.data(), sizeof(int) * P.n); }
void ppsim_get_levels(void* h, int* lv) { Plan& P = *(Plan*)h; std::memcpy(lv, P.piv_level.data(), sizeof(int) * P.npiv); }
void ppsim_get_level_task_counts(void* h, int* cnt) {
  Plan& P = *(Plan*)h;
  for (int l = 0; l < P.n_levels; ++l) cnt[l] = P.flevel_ptr[l + 1] - P.flevel_ptr[l];
}

// per gather/fused task: level, pivot width, rows, kind, number of entries
void ppsim_task_profile(void* h, int* out /*5 per task*/) {
  Plan& P = *(Plan*)h;
  for (size_t t = 0; t < P.ftasks.size(); ++t) {
    const auto& ft = P.ftasks[t];
    if (ft.kind < 0) { out[5 * t] = -1; out[5 * t + 1] = 0; out[5 * t + 2] = 0; out[5 * t + 3] = -1; out[5 * t + 4] = 0; continue; }
    const int w = P.piv_w[ft.piv], nrow = ft.r1 - ft.r0;
    out[5 * t] = P.piv_level[ft.piv]; out[5 * t + 1] = w; out[5 * t + 2] = nrow;
    out[5 * t + 3] = ft.kind; out[5 * t + 4] = P.fdst_ptr[ft.dptr0 + nrow] - P.fdst_ptr[ft.dptr0];
  }
}

// per level (8 ints): fused tasks, gather chunks (pieces counted), split rows, scale tasks, big panels, rows of the big
// panels (sum), rows of the tallest big panel, longest gather/fused task (entries)
void ppsim_level_profile(void* h, int* out) {
  Plan& P = *(Plan*)h;
  for (int l = 0; l < P.n_levels; ++l) {
    int* o = out + 8 * l;
    for (int i = 0; i < 8; ++i) o[i] = 0;
    int lastp = -1;
    for (int t = P.flevel_ptr[l]; t < P.flevel_ptr[l + 1]; ++t) {
      const auto& ft = P.ftasks[t];
      if (ft.kind < 0) continue;
      if (ft.kind == 1) o[0]++;
      else {
        o[1]++;
        if (ft.npieces > 1 && ft.piece == 0) o[2]++;
      }
    }
    for (int t = P.slevel_ptr[l]; t < P.slevel_ptr[l + 1]; ++t) {
      const auto& st = P.stasks[t];
      o[3]++;
      if (st.piv != lastp) {
        lastp = st.piv;
        const int f = P.piv_w[st.piv] + P.piv_rowptr[st.piv + 1] - P.piv_rowptr[st.piv];
        o[4]++; o[5] += f; o[6] = std::max(o[6], f);
      }
    }
    o[7] = P.flevel_maxent[l];
  }
}

// per level (4 ints): most gather chunks (pieces counted) of one big panel, most rows of one gather chunk, big panels
// with more than 4 / more than 8 gather chunks
void ppsim_level_teams(void* h, int* out) {
  Plan& P = *(Plan*)h;
  for (int l = 0; l < P.n_levels; ++l) {
    int* o = out + 4 * l;
    o[0] = o[1] = o[2] = o[3] = 0;
    std::vector<int> cnt(P.npiv, 0);
    for (int t = P.flevel_ptr[l]; t < P.flevel_ptr[l + 1]; ++t) {
      const auto& ft = P.ftasks[t];
      if (ft.kind != 0) continue;
      cnt[ft.piv]++;
      o[1] = std::max(o[1], ft.r1 - ft.r0);
    }
    for (int p = 0; p < P.npiv; ++p) { o[0] = std::max(o[0], cnt[p]); o[2] += cnt[p] > 4; o[3] += cnt[p] > 8; }
  }
}

// Operand loads of the gather tasks per level (3 int64): as scheduled (1 U + w L per product entry), and if the rows of a
// panel were gathered in groups of up to 4 consecutive rows sharing the L operands of a source column (w L per distinct
// source column of the group + 1 U per row that has it); third value: product entries.  Diagnostic.
void ppsim_operand_loads(void* h, long long* out) {
  Plan& P = *(Plan*)h;
  for (int l = 0; l < P.n_levels; ++l) {
    long long* o = out + 3 * l;
    o[0] = o[1] = o[2] = 0;
    // rows of one pivot in this level, by row slot: list of L positions (the source column identity)
    std::map<std::pair<int, int>, std::vector<int>> rows;      // (pivot, row slot) -> l of its product entries
    for (int t = P.flevel_ptr[l]; t < P.flevel_ptr[l + 1]; ++t) {
      const auto& ft = P.ftasks[t];
      if (ft.kind < 0) continue;
      const int w = P.piv_w[ft.piv];
      if (ft.npieces > 1) {
        auto& v = rows[{ft.piv, ft.r0}];
        for (int e = P.fdst_ptr[ft.dptr0]; e < P.fdst_ptr[ft.dptr0 + 1]; ++e)
          if (P.fentries[e].u >= 0) { v.push_back(P.fentries[e].l); o[0] += 1 + w; o[2] += 1; }
        continue;
      }
      for (int rr = 0; rr < ft.r1 - ft.r0; ++rr) {
        auto& v = rows[{ft.piv, ft.r0 + rr}];
        for (int e = P.fdst_ptr[ft.dptr0 + rr]; e < P.fdst_ptr[ft.dptr0 + rr + 1]; ++e)
          if (P.fentries[e].u >= 0) { v.push_back(P.fentries[e].l); o[0] += 1 + w; o[2] += 1; }
      }
    }
    auto it = rows.begin();
    while (it != rows.end()) {
      const int piv = it->first.first, w = P.piv_w[piv];
      std::map<int, int> have;                                   // l -> rows of the group that have it
      int n = 0;
      int last_slot = it->first.second - 1;
      while (it != rows.end() && it->first.first == piv && n < 4 && it->first.second >= w && it->first.second == last_slot + 1) {
        for (int lpos : it->second) have[lpos]++;
        last_slot = it->first.second;
        ++n; ++it;
      }
      if (n == 0) {                                              // a pivot-block row (kept as it is) or a gap: alone
        for (int lpos : it->second) { (void)lpos; o[1] += 1 + w; }
        ++it;
        continue;
      }
      for (auto& kv : have) o[1] += w + kv.second;
    }
  }
}

// per pivot: level and width (diagnostics)
void ppsim_get_piv_level(void* h, int* level, int* width) {
  Plan& P = *(Plan*)h;
  std::memcpy(level, P.piv_level.data(), sizeof(int) * P.npiv);
  std::memcpy(width, P.piv_w.data(), sizeof(int) * P.npiv);
}

// per pivot: number of coupling rows in its panel
void ppsim_get_ncrow(void* h, int* out) { Plan& P = *(Plan*)h; std::memcpy(out, P.piv_ncrow.data(), sizeof(int) * P.npiv); }

// solve schedule per level (4 ints): scalar columns, entries of the longest / of all forward rows, rows of the longest column
void ppsim_solve_levels(void* h, int* out) {
  Plan& P = *(Plan*)h;
  for (int l = 0; l < P.n_levels; ++l) {
    int* o = out + 4 * l;
    o[0] = P.clevel_ptr[l + 1] - P.clevel_ptr[l]; o[1] = o[2] = o[3] = 0;
    for (int q = P.clevel_ptr[l]; q < P.clevel_ptr[l + 1]; ++q) {
      const int c = P.clevel_col[q], pv = P.piv_of_col[c];
      const int ne = P.sfwd_eptr[c + 1] - P.sfwd_eptr[c];
      o[1] = std::max(o[1], ne); o[2] += ne;
      o[3] = std::max(o[3], P.piv_rowptr[pv + 1] - P.piv_rowptr[pv]);
    }
  }
}

namespace {
// rows [r0, r1) of panel p: L = U inv(P)
void scale_rows(const Plan& P, int p, int r0, int r1, const double* inv, const double* U, double* L) {
  const int w = P.piv_w[p];
  for (int r = std::max(r0, w); r < r1; ++r) {
    const double* u = &U[P.piv_uoff[p] + (int64_t)r * w];
    double* l = &L[P.piv_uoff[p] + (int64_t)r * w];
    for (int t2 = 0; t2 < w; ++t2) {
      double v = 0.0;
      for (int t1 = 0; t1 < w; ++t1) {
        const int hi = t1 > t2 ? t1 : t2, lo = t1 > t2 ? t2 : t1;
        v += u[t1] * inv[hi * (hi + 1) / 2 + lo];
      }
      l[t2] = v;
      if (g_growth_bound > 0.0 && std::fabs(v) > g_growth_bound) g_last_growth = 1;
    }
  }
}
}  // namespace

// One instance.  can: canonical values (ncan).  U, L: usize, Dinv: dsize, S: nc*nc (row-major, lower filled;
// contribution -A K^-1 A^T), inertia[3] += (pos, neg, zero).
int ppsim_factor(void* h, const double* can, double* U, double* L, double* Dinv, double* S, int64_t* inertia,
                 double eps) {
  Plan& P = *(Plan*)h;
  g_last_growth = 0;
  g_first_zero_piv = -1;
  std::memset(U, 0, sizeof(double) * P.usize);
  std::memset(L, 0, sizeof(double) * P.usize);
  std::vector<double> Tm((size_t)std::max(P.bsize, 1), 0.0);
  int pos = 0, neg = 0, zero = 0;
  for (int lvl = 0; lvl < P.n_levels; ++lvl) {
    // launch G: gather chunks and fused small panels
    for (int ti = P.flevel_ptr[lvl]; ti < P.flevel_ptr[lvl + 1]; ++ti) {
      const auto& t = P.ftasks[ti];
      if (t.kind < 0) continue;                       // quad padding
      const int p = t.piv, wp = P.piv_w[p], w = t.ws > 0 ? t.ws : wp, qoff = t.qoff;   // w: columns of the task's slice
      const int nrow = t.r1 - t.r0;
      double blk[PP_WMAX * PP_WMAX] = {0}, tmd[PP_WMAX] = {0}, inv[PP_WMAX * (PP_WMAX + 1) / 2] = {0};
      if (t.npieces > 1) {
        // one long row gathered by the waves of a quad: partial sums per piece, added in piece order by piece 0
        if (t.piece != 0) continue;                   // (the pieces follow piece 0 in the task list)
        double acc[PP_WMAX] = {0}, tmax[PP_WMAX] = {0};
        for (int j = 0; j < t.npieces; ++j) {
          const auto& tj = P.ftasks[ti + j];
          double pa[PP_WMAX] = {0}, pm[PP_WMAX] = {0};
          for (int e = P.fdst_ptr[tj.dptr0]; e < P.fdst_ptr[tj.dptr0 + 1]; ++e) {
            const auto& fe = P.fentries[e];
            if (fe.u < 0) {
              const double v = can[-1 - fe.u];
              pa[fe.q] += v;
              pm[fe.q] = std::fmax(pm[fe.q], std::fabs(v));
            } else {
              const double su = U[fe.u];
              const int q0 = fe.q & 15, qn = (fe.q >> 4) & 15;      // a run of columns of the destination row
              for (int j = 0; j < qn; ++j) {
                const double term = su * L[fe.l + j * fe.wk];
                pa[q0 + j] -= term;
                pm[q0 + j] = std::fmax(pm[q0 + j], std::fabs(term));
              }
            }
          }
          for (int q = 0; q < w; ++q) { acc[q] = (j == 0) ? pa[q] : acc[q] + pa[q]; tmax[q] = std::fmax(tmax[q], pm[q]); }
        }
        for (int q = 0; q < w; ++q) {
          U[P.piv_uoff[p] + (int64_t)t.r0 * wp + qoff + q] = acc[q];
          if (t.r0 < wp) Tm[P.piv_boff[p] + (t.r0 * wp + qoff + q)] = tmax[q];
        }
        continue;
      }
      for (int rr = 0; rr < nrow; ++rr) {
        const int slot = t.r0 + rr;
        double acc[PP_WMAX] = {0}, tmax[PP_WMAX] = {0};
        for (int e = P.fdst_ptr[t.dptr0 + rr]; e < P.fdst_ptr[t.dptr0 + rr + 1]; ++e) {
          const auto& fe = P.fentries[e];
          if (fe.u < 0) {
            const double v = can[-1 - fe.u];
            acc[fe.q] += v;
            tmax[fe.q] = std::fmax(tmax[fe.q], std::fabs(v));
          } else {
            const double su = U[fe.u];
            const int q0 = fe.q & 15, qn = (fe.q >> 4) & 15;      // a run of columns of the destination row
            for (int j = 0; j < qn; ++j) {
              const double term = su * L[fe.l + j * fe.wk];
              acc[q0 + j] -= term;
              tmax[q0 + j] = std::fmax(tmax[q0 + j], std::fabs(term));
            }
          }
        }
        for (int q = 0; q < w; ++q) {
          U[P.piv_uoff[p] + (int64_t)slot * wp + qoff + q] = acc[q];
          if (slot < wp) {
            if (t.kind == 0) Tm[P.piv_boff[p] + (slot * wp + qoff + q)] = tmax[q];
            else { blk[slot * PP_WMAX + q] = acc[q]; if (q == slot) tmd[slot] = tmax[q]; }   // (term magnitudes of the diagonal entries)
          }
        }
        if (t.kind == 1 && slot == w - 1) {
          const int code = pp::invert_block(w, P.piv_sub[p], blk, tmd, eps, inv);
          for (int q = 0; q < w * (w + 1) / 2; ++q) Dinv[P.piv_doff[p] + q] = inv[q];
          pos += code & 15; neg += (code >> 4) & 15; zero += (code >> 8) & 15;
          if (((code >> 8) & 15) && g_first_zero_piv < 0) g_first_zero_piv = p;
        }
      }
      if (t.kind == 1) scale_rows(P, p, t.r0, t.r1, &Dinv[P.piv_doff[p]], U, L);
    }
    // tile tasks (k_gather_tiles): up to PP_TILE_ROWS rows of a panel, cut into pieces by source panels; partial sums per
    // piece, added in piece order
    for (int ti = P.tlevel_ptr[lvl]; ti < P.tlevel_ptr[lvl + 1]; ++ti) {
      const auto& t = P.ttasks[ti];
      if (t.kind < 0 || t.piece != 0) continue;
      const int p = t.piv, wp = P.piv_w[p], w = t.ws > 0 ? t.ws : wp, qoff = t.qoff;
      const int nrow = t.r1 - t.r0;
      double acc[PP_TILE_ROWS][PP_WMAX] = {{0}}, tmax[PP_TILE_ROWS][PP_WMAX] = {{0}};
      for (int j = 0; j < t.npieces; ++j) {
        const auto& tj = P.ttasks[ti + j];
        for (int rr = 0; rr < nrow; ++rr) {
          double pa[PP_WMAX] = {0}, pm[PP_WMAX] = {0};
          for (int e = P.fdst_ptr[tj.dptr0 + rr]; e < P.fdst_ptr[tj.dptr0 + rr + 1]; ++e) {
            const auto& fe = P.fentries[e];
            if (fe.u < 0) {
              const double v = can[-1 - fe.u];
              pa[fe.q] += v;
              pm[fe.q] = std::fmax(pm[fe.q], std::fabs(v));
            } else {
              const double su = U[fe.u];
              const int q0 = fe.q & 15, qn = (fe.q >> 4) & 15;
              for (int jj = 0; jj < qn; ++jj) {
                const double term = su * L[fe.l + jj * fe.wk];
                pa[q0 + jj] -= term;
                pm[q0 + jj] = std::fmax(pm[q0 + jj], std::fabs(term));
              }
            }
          }
          for (int q = 0; q < w; ++q) { acc[rr][q] = (j == 0) ? pa[q] : acc[rr][q] + pa[q]; tmax[rr][q] = std::fmax(tmax[rr][q], pm[q]); }
        }
      }
      for (int rr = 0; rr < nrow; ++rr)
        for (int q = 0; q < w; ++q) {
          const int slot = t.r0 + rr;
          U[P.piv_uoff[p] + (int64_t)slot * wp + qoff + q] = acc[rr][q];
          if (slot < wp) Tm[P.piv_boff[p] + (slot * wp + qoff + q)] = tmax[rr][q];
        }
    }
    // launch S: scale chunks of big panels
    for (int ti = P.slevel_ptr[lvl]; ti < P.slevel_ptr[lvl + 1]; ++ti) {
      const auto& t = P.stasks[ti];
      const int p = t.piv, w = P.piv_w[p];
      double blk[PP_WMAX * PP_WMAX] = {0}, tmd[PP_WMAX] = {0}, inv[PP_WMAX * (PP_WMAX + 1) / 2] = {0};
      for (int q = 0; q < w * w; ++q) {
        blk[(q / w) * PP_WMAX + q % w] = U[P.piv_uoff[p] + q];
        if (q / w == q % w) tmd[q / w] = Tm[P.piv_boff[p] + q];
      }
      const int code = pp::invert_block(w, P.piv_sub[p], blk, tmd, eps, inv);
      if (t.r0 == w) {
        for (int q = 0; q < w * (w + 1) / 2; ++q) Dinv[P.piv_doff[p] + q] = inv[q];
        pos += code & 15; neg += (code >> 4) & 15; zero += (code >> 8) & 15;
          if (((code >> 8) & 15) && g_first_zero_piv < 0) g_first_zero_piv = p;
      }
      scale_rows(P, p, t.r0, t.r1, inv, U, L);
    }
    // chain fronts of this level (k_chain_front): the panels of a front one after the other -- block inversion, scaling,
    // update of the later panels' gathered values by this panel (every row of a later panel is a row of this one)
    for (int c = P.chain_lvl_ptr[lvl]; c < P.chain_lvl_ptr[lvl + 1]; ++c) {
      for (int ti = P.chain_ptr[c]; ti < P.chain_ptr[c + 1]; ++ti) {
        const int p = P.chain_piv[ti], w = P.piv_w[p], c0 = P.chain_col0[ti];
        const int f = w + (P.piv_rowptr[p + 1] - P.piv_rowptr[p]);
        double blk[PP_WMAX * PP_WMAX] = {0}, tmd[PP_WMAX] = {0}, inv[PP_WMAX * (PP_WMAX + 1) / 2] = {0};
        for (int q = 0; q < w * w; ++q) {
          blk[(q / w) * PP_WMAX + q % w] = U[P.piv_uoff[p] + q];
          if (q / w == q % w) tmd[q / w] = Tm[P.piv_boff[p] + q];
        }
        const int code = pp::invert_block(w, P.piv_sub[p], blk, tmd, eps, inv);
        for (int q = 0; q < w * (w + 1) / 2; ++q) Dinv[P.piv_doff[p] + q] = inv[q];
        pos += code & 15; neg += (code >> 4) & 15; zero += (code >> 8) & 15;
        if (((code >> 8) & 15) && g_first_zero_piv < 0) g_first_zero_piv = p;
        scale_rows(P, p, w, f, inv, U, L);
        for (int tj = ti + 1; tj < P.chain_ptr[c + 1]; ++tj) {
          const int pj = P.chain_piv[tj], wj = P.piv_w[pj], cj = P.chain_col0[tj];
          const int fj = wj + (P.piv_rowptr[pj + 1] - P.piv_rowptr[pj]);
          for (int sl = 0; sl < fj; ++sl)
            for (int q = 0; q < wj; ++q) {
              const int tr = cj + sl - c0, tc = cj + q - c0;       // slots of the row and of the column in panel p
              double acc = U[P.piv_uoff[pj] + (int64_t)sl * wj + q];
              for (int k = 0; k < w; ++k) {
                const double term = U[P.piv_uoff[p] + (int64_t)tr * w + k] * L[P.piv_uoff[p] + (int64_t)tc * w + k];
                acc -= term;
                if (sl < wj && q == sl) Tm[P.piv_boff[pj] + sl * wj + sl] = std::fmax(Tm[P.piv_boff[pj] + sl * wj + sl], std::fabs(term));
              }
              U[P.piv_uoff[pj] + (int64_t)sl * wj + q] = acc;
            }
        }
      }
    }
    // root front: inversion of its gathered pivot block (k_front_invert), rows scaled with the explicit inverse (k_scale_wide)
    if (P.front_piv >= 0 && P.piv_flevel[P.front_piv] == lvl) {
      const int p = P.front_piv, w = P.piv_w[p];
      double A[pp::PP_WF * pp::PP_WF] = {0}, tmd[pp::PP_WF] = {0}, finv[pp::PP_WF * (pp::PP_WF + 1) / 2] = {0};
      for (int i = 0; i < w; ++i)
        for (int j = 0; j < w; ++j) {
          const int hi = i > j ? i : j, lo = i > j ? j : i;       // (the lower triangle of the gathered block is the matrix)
          A[i * pp::PP_WF + j] = U[P.piv_uoff[p] + (int64_t)hi * w + lo];
          if (i == j) tmd[i] = Tm[P.piv_boff[p] + i * w + i];
        }
      const int code = pp::invert_front(w, P.piv_sub[p], A, tmd, eps, finv);
      for (int q = 0; q < w * (w + 1) / 2; ++q) Dinv[P.piv_doff[p] + q] = finv[q];
      pos += code & 15; neg += (code >> 4) & 15; zero += (code >> 8) & 15;
          if (((code >> 8) & 15) && g_first_zero_piv < 0) g_first_zero_piv = p;
      for (auto& t : P.wtasks) scale_rows(P, p, t.r0, t.r1, finv, U, L);
    }
  }
  inertia[0] += pos; inertia[1] += neg; inertia[2] += zero;
  // Schur tiles S -= L_c,a U_c,b^T over the coupling rows of every panel
  const int T = P.opt.tile, nc = P.nc;
  for (size_t ti = 0; ti < P.stile_a.size(); ++ti) {
    double accS[8][8] = {};
    for (int r = P.stile_ptr[ti]; r < P.stile_ptr[ti + 1]; ++r) {
      const auto& rec = P.stile_rec[r];
      const int p = rec.piv, w = P.piv_w[p];
      for (int i = 0; i < T; ++i) {
        if (rec.slotA[i] < 0) continue;
        const double* la = &L[P.piv_uoff[p] + (int64_t)rec.slotA[i] * w];
        for (int j = 0; j < T; ++j) {
          if (rec.slotB[j] < 0) continue;
          const double* ub = &U[P.piv_uoff[p] + (int64_t)rec.slotB[j] * w];
          double v = 0.0;
          for (int t2 = 0; t2 < w; ++t2) v += la[t2] * ub[t2];
          accS[i][j] -= v;
        }
      }
    }
    for (int i = 0; i < T; ++i)
      for (int j = 0; j < T; ++j) {
        int ci = P.stile_a[ti] * T + i, cj = P.stile_b[ti] * T + j;
        if (ci < nc && cj < nc && ci >= cj) S[(size_t)ci * nc + cj] += accS[i][j];
      }
  }
  return zero > 0 ? 2 : 0;
}

// forward: Y (n+nc): Y[c] = b_c - sum L[c,k] Y[k] for c < n in level order; Y[n+c] = -A K^-1 r contribution
void ppsim_forward(void* h, const double* L, const double* rhs, double* Y) {
  Plan& P = *(Plan*)h;
  for (int i = 0; i < P.n; ++i) {
    const int c = P.clevel_col[i];
    double acc = rhs[P.perm[c]];
    for (int e = P.sfwd_eptr[c]; e < P.sfwd_eptr[c + 1]; ++e) acc -= L[P.sfwd_upos[e]] * Y[P.sfwd_zcol[e]];
    const int pv = P.piv_of_col[c], fr = P.piv_chain[pv];
    if (P.chain_sweeps_on && fr >= 0) {
      // (k_chain_fwd: the earlier panels of the front, in order; column c is row c0_j + q - c0_i of panel i)
      int tj = P.chain_ptr[fr];
      while (P.chain_piv[tj] != pv) ++tj;
      const int cf = P.chain_col0[tj] + (c - P.piv_start[pv]);
      for (int ti = P.chain_ptr[fr]; ti < tj; ++ti) {
        const int pi = P.chain_piv[ti], wi = P.piv_w[pi];
        const int64_t row = P.piv_uoff[pi] + (int64_t)(cf - P.chain_col0[ti]) * wi;
        for (int k = 0; k < wi; ++k) acc -= L[row + k] * Y[P.piv_start[pi] + k];
      }
    }
    Y[c] = acc;
  }
  for (int c = 0; c < P.nc; ++c) {
    double s = 0;
    for (int e = P.crow_eptr[c]; e < P.crow_eptr[c + 1]; ++e) s -= L[P.crow_upos[e]] * Y[P.crow_zcol[e]];
    Y[P.n + c] = s;
  }
}

// backward: X[c] = (inv(P) y_p)_q - sum_i L[i, c] X[row i]; X[n..] = x_c on entry; x (original order) on exit
void ppsim_backward(void* h, const double* L, const double* Dinv, const double* Y, double* X, double* x) {
  Plan& P = *(Plan*)h;
  for (int i = P.n - 1; i >= 0; --i) {
    const int c = P.clevel_col[i];
    const int p = P.piv_of_col[c], w = P.piv_w[p], p0 = P.piv_start[p], q = c - p0;
    const double* inv = &Dinv[P.piv_doff[p]];
    double z = 0.0;
    for (int t = 0; t < w; ++t) { const int hi = q > t ? q : t, lo = q > t ? t : q; z += inv[hi * (hi + 1) / 2 + lo] * Y[p0 + t]; }
    const int nr = P.piv_rowptr[p + 1] - P.piv_rowptr[p];
    const int* ri = &P.rowidx[P.piv_rowptr[p]];
    const double* l = &L[P.piv_uoff[p] + (int64_t)w * w];
    double g = 0.0;
    for (int j = 0; j < nr; ++j) g += l[(int64_t)j * w + q] * X[ri[j]];
    X[c] = z - g;
  }
  for (int k = 0; k < P.n; ++k) x[P.perm[k]] = X[k];
}

// block pivot inversion by static-order sweeps (pivot.hpp): returns the code, inv packed lower
int ppsim_invert_block(int w, unsigned sub, const double* a, double colmax, double eps, double* inv) {
  double blk[PP_WMAX * PP_WMAX] = {0};
  for (int i = 0; i < w; ++i)
    for (int j = 0; j < w; ++j) blk[i * PP_WMAX + j] = a[i * w + j];
  double tmd[PP_WMAX];
  for (int i = 0; i < PP_WMAX; ++i) tmd[i] = colmax;      // (one bound for every diagonal entry)
  return pp::invert_block(w, sub, blk, tmd, eps, inv);
}

// root front: static-order sweeps on a w x w block (row-major, stride w; lower triangle read): code, inv packed lower
int ppsim_invert_front(int w, unsigned sub, const double* a, double colmax, double eps, double* inv) {
  double A[pp::PP_WF * pp::PP_WF] = {0};
  for (int i = 0; i < w; ++i)
    for (int j = 0; j < w; ++j) { const int hi = i > j ? i : j, lo = i > j ? j : i; A[i * pp::PP_WF + j] = a[hi * w + lo]; }
  double tmd[pp::PP_WF];
  for (int i = 0; i < pp::PP_WF; ++i) tmd[i] = colmax;
  return pp::invert_front(w, sub, A, tmd, eps, inv);
}

// dense Bunch-Kaufman on a column-major n x n matrix (lower triangle read); info = (pos, neg, zero)
void ppsim_bk_factor(int n, double* A, int* ipiv, int* info, double eps) {
  HostCtx ctx;
  std::vector<double> work(2 * (size_t)n);
  pp::BkInfo bi;
  pp::bk_factor(ctx, n, A, n, ipiv, work.data(), &bi, eps);
  info[0] = bi.npos; info[1] = bi.nneg; info[2] = bi.nzero;
}
void ppsim_bk_solve(int n, const double* A, const int* ipiv, double* b) {
  HostCtx ctx;
  pp::bk_solve(ctx, n, A, n, ipiv, b);
}

}  // extern "C"

extern "C" {
// per level (4 int64): product entries over all columns of the destination, over a shorter run of columns, initial-value
// entries, columns covered by the shorter runs -- diagnostic
void ppsim_entry_kinds(void* h, long long* out) {
  Plan& P = *(Plan*)h;
  for (int l = 0; l < P.n_levels; ++l) {
    long long* o = out + 4 * l;
    o[0] = o[1] = o[2] = o[3] = 0;
    for (int t = P.flevel_ptr[l]; t < P.flevel_ptr[l + 1]; ++t) {
      const auto& ft = P.ftasks[t];
      if (ft.kind < 0) continue;
      const int nrow = (ft.npieces > 1) ? 1 : ft.r1 - ft.r0;
      for (int rr = 0; rr < nrow; ++rr)
        for (int e = P.fdst_ptr[ft.dptr0 + rr]; e < P.fdst_ptr[ft.dptr0 + rr + 1]; ++e) {
          const auto& fe = P.fentries[e];
          if (fe.u < 0) o[2]++;
          else if (((fe.q >> 4) & 15) == P.piv_w[ft.piv] || (ft.ws > 0 && ((fe.q >> 4) & 15) == ft.ws)) o[0]++;
          else { o[1]++; o[3] += (fe.q >> 4) & 15; }
        }
    }
  }
}
}
