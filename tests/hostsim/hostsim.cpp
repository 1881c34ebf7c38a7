// TEST-ONLY host interpreter of the symbolic plan (never shipped, never a fallback).
// Executes exactly the task records the HIP kernels execute (same order, same
// formulas, one instance at a time) so the schedule can be validated without a GPU.
#include <cmath>
#include <cstdint>
#include <cstring>
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

static int g_sn_wmax = 0, g_sn_tol = -1;

extern "C" {

// test knob: supernode width cap / padded-row tolerance for plans created afterwards (0 / -1: defaults)
void ppsim_set_supernodes(int wmax, int tol) { g_sn_wmax = wmax; g_sn_tol = tol; }

void* ppsim_create(int n, int nc, int nnzK, const int* rowK, const int* colK, int nnzB, const int* rowB,
                   const int* colB, const double* vals, int max_entries, int delta_abs, double delta_rel) {
  auto* P = new Plan();
  pp::PlanOptions opt;
  if (max_entries > 0) opt.max_task_entries = max_entries;
  if (delta_abs >= 0) opt.md_delta_abs = delta_abs;
  if (delta_rel >= 0) opt.md_delta_rel = delta_rel;
  if (g_sn_wmax > 0) opt.sn_wmax = g_sn_wmax;
  if (g_sn_tol >= 0) opt.sn_tol_rows = g_sn_tol;
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
void ppsim_get_perm(void* h, int* perm) { Plan& P = *(Plan*)h; std::memcpy(perm, P.perm.data(), sizeof(int) * P.n); }
void ppsim_get_levels(void* h, int* lv) { Plan& P = *(Plan*)h; std::memcpy(lv, P.piv_level.data(), sizeof(int) * P.npiv); }
void ppsim_get_level_task_counts(void* h, int* cnt) {
  Plan& P = *(Plan*)h;
  for (int l = 0; l < P.n_levels; ++l) cnt[l] = P.flevel_ptr[l + 1] - P.flevel_ptr[l];
}

// per task: level, pivot width, rows, number of multiplier scalars, number of entries
void ppsim_task_profile(void* h, int* out /*5 per task*/) {
  Plan& P = *(Plan*)h;
  for (size_t t = 0; t < P.ftasks.size(); ++t) {
    const auto& ft = P.ftasks[t];
    const int w = P.piv_w[ft.piv], ndst = (ft.r1 - ft.r0) * w;
    out[5 * t] = P.piv_level[ft.piv]; out[5 * t + 1] = w; out[5 * t + 2] = ft.r1 - ft.r0;
    out[5 * t + 3] = ft.m1 - ft.m0; out[5 * t + 4] = P.fdst_ptr[ft.dptr0 + ndst] - P.fdst_ptr[ft.dptr0];
  }
}

// One instance.  can: canonical values (ncan).  U: usize, Dinv: 3*npiv, S: nc*nc (row-major,
// lower filled; contribution -A K^-1 A^T), inertia[3] += (pos, neg, zero).
int ppsim_factor(void* h, const double* can, double* U, double* Dinv, double* S, int64_t* inertia, double eps) {
  Plan& P = *(Plan*)h;
  std::memset(U, 0, sizeof(double) * P.usize);
  std::vector<double> M;
  int pos = 0, neg = 0, zero = 0;
  for (const auto& t : P.ftasks) {
    const int p = t.piv, w = P.piv_w[p];
    const int64_t dst0 = P.piv_uoff[p] + (int64_t)t.r0 * w;
    const int ndst = (t.r1 - t.r0) * w;
    M.assign(1 + (t.m1 - t.m0), -1.0);
    for (int j = t.m0; j < t.m1; ++j) {
      const auto& m = P.mrecs[j];
      double v = 0.0;
      for (int q = 0; q < PP_WMAX; ++q) if (m.d[q] >= 0) v += Dinv[m.d[q]] * U[m.u[q]];
      M[1 + (j - t.m0)] = v;
    }
    double piv[PP_WMAX * PP_WMAX] = {0}, tmax_diag = 0.0, colmax = 0.0;
    for (int d = 0; d < ndst; ++d) {
      double acc = 0.0, tmax = 0.0;
      for (int e = P.fdst_ptr[t.dptr0 + d]; e < P.fdst_ptr[t.dptr0 + d + 1]; ++e) {
        const auto& fe = P.fentries[e];
        const double src = (fe.src >= 0) ? U[fe.src] : can[-1 - fe.src];
        const double term = src * M[fe.midx];
        acc -= term;
        tmax = std::fmax(tmax, std::fabs(term));
      }
      U[dst0 + d] = acc;
      if (t.r0 == 0 && d < w * w) { piv[d] = acc; tmax_diag = std::fmax(tmax_diag, tmax); }
      else colmax = std::fmax(colmax, std::fabs(acc));
    }
    if (t.r0 == 0) {
      const int code = pp::invert_block(w, P.piv_sub[p], piv, std::fmax(colmax, tmax_diag), eps, &Dinv[P.piv_doff[p]]);
      pos += code & 15; neg += (code >> 4) & 15; zero += (code >> 8) & 15;
    }
  }
  inertia[0] += pos; inertia[1] += neg; inertia[2] += zero;
  // scaled coupling rows Lc = U_c inv(P), then Schur tiles S -= Lc_a U_c_b^T
  std::vector<double> Lc((size_t)P.lcsize, 0.0);
  for (int p = 0; p < P.npiv; ++p) {
    if (P.piv_lcoff[p] < 0) continue;
    const int w = P.piv_w[p];
    const double* inv = &Dinv[P.piv_doff[p]];
    for (int r = 0; r < P.piv_ncrow[p]; ++r) {
      const double* u = &U[P.piv_uoff[p] + (int64_t)(P.piv_cslot0[p] + r) * w];
      for (int t2 = 0; t2 < w; ++t2) {
        double v = 0.0;
        for (int t1 = 0; t1 < w; ++t1) {
          const int hi = t1 > t2 ? t1 : t2, lo = t1 > t2 ? t2 : t1;
          v += u[t1] * inv[hi * (hi + 1) / 2 + lo];
        }
        Lc[(size_t)P.piv_lcoff[p] + (size_t)r * w + t2] = v;
      }
    }
  }
  const int T = P.opt.tile, nc = P.nc;
  for (size_t ti = 0; ti < P.stile_a.size(); ++ti) {
    double accS[8][8] = {};
    for (int r = P.stile_ptr[ti]; r < P.stile_ptr[ti + 1]; ++r) {
      const auto& rec = P.stile_rec[r];
      const int p = rec.piv, w = P.piv_w[p];
      for (int i = 0; i < T; ++i) {
        if (rec.slotA[i] < 0) continue;
        const double* la = &Lc[(size_t)P.piv_lcoff[p] + (size_t)(rec.slotA[i] - P.piv_cslot0[p]) * w];
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

// forward: W (n+nc) gets permuted rhs in [0,n); on exit W[0,n) = z, W[n+c] = -A K^-1 r contribution
void ppsim_forward(void* h, const double* U, const double* Dinv, const double* rhs, double* W) {
  Plan& P = *(Plan*)h;
  for (int li = 0; li < P.npiv; ++li) {
    const int p = P.lvl_piv[li], w = P.piv_w[p], p0 = P.piv_start[p];
    double y[PP_WMAX] = {0};
    for (int q = 0; q < w; ++q) {
      double acc = rhs[P.perm[p0 + q]];
      for (int e = P.sfwd_eptr[p0 + q]; e < P.sfwd_eptr[p0 + q + 1]; ++e) acc -= U[P.sfwd_upos[e]] * W[P.sfwd_zcol[e]];
      y[q] = acc;
    }
    const double* inv = &Dinv[P.piv_doff[p]];
    for (int q = 0; q < w; ++q) {
      double z = 0.0;
      for (int t = 0; t < w; ++t) { const int hi = q > t ? q : t, lo = q > t ? t : q; z += inv[hi * (hi + 1) / 2 + lo] * y[t]; }
      W[p0 + q] = z;
    }
  }
  for (int c = 0; c < P.nc; ++c) {
    double s = 0;
    for (int e = P.crow_eptr[c]; e < P.crow_eptr[c + 1]; ++e) s -= U[P.crow_upos[e]] * W[P.crow_zcol[e]];
    W[P.n + c] = s;
  }
}

// backward: W[0,n) = z, W[n..] = x_c on entry; x (original order) on exit
void ppsim_backward(void* h, const double* U, const double* Dinv, double* W, double* x) {
  Plan& P = *(Plan*)h;
  for (int li = P.npiv - 1; li >= 0; --li) {
    const int p = P.lvl_piv[li], w = P.piv_w[p], p0 = P.piv_start[p];
    double g[PP_WMAX] = {0};
    const int nr = P.piv_rowptr[p + 1] - P.piv_rowptr[p];
    const int* ri = &P.rowidx[P.piv_rowptr[p]];
    const double* u = &U[P.piv_uoff[p] + (int64_t)w * w];
    for (int j = 0; j < nr; ++j)
      for (int q = 0; q < w; ++q) g[q] += u[(int64_t)j * w + q] * W[ri[j]];
    const double* inv = &Dinv[P.piv_doff[p]];
    for (int q = 0; q < w; ++q) {
      double z = 0.0;
      for (int t = 0; t < w; ++t) { const int hi = q > t ? q : t, lo = q > t ? t : q; z += inv[hi * (hi + 1) / 2 + lo] * g[t]; }
      W[p0 + q] -= z;
    }
  }
  for (int k = 0; k < P.n; ++k) x[P.perm[k]] = W[k];
}

// block pivot inversion by static-order sweeps (pivot.hpp): returns the code, inv packed lower
int ppsim_invert_block(int w, unsigned sub, const double* a, double colmax, double eps, double* inv) {
  return pp::invert_block(w, sub, a, colmax, eps, inv);
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
