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

extern "C" {

void* ppsim_create(int n, int nc, int nnzK, const int* rowK, const int* colK, int nnzB, const int* rowB,
                   const int* colB, const double* vals, int acc_doubles, int delta_abs, double delta_rel) {
  auto* P = new Plan();
  pp::PlanOptions opt;
  if (acc_doubles > 0) opt.acc_doubles = acc_doubles;
  if (delta_abs >= 0) opt.md_delta_abs = delta_abs;
  if (delta_rel >= 0) opt.md_delta_rel = delta_rel;
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
  out[10] = (int64_t)P.runs.size(); out[11] = (int64_t)P.stile_a.size(); out[12] = (int64_t)P.stile_rec.size();
}
void ppsim_get_perm(void* h, int* perm) { Plan& P = *(Plan*)h; std::memcpy(perm, P.perm.data(), sizeof(int) * P.n); }
void ppsim_get_levels(void* h, int* lv) { Plan& P = *(Plan*)h; std::memcpy(lv, P.piv_level.data(), sizeof(int) * P.npiv); }
void ppsim_get_level_task_counts(void* h, int* cnt) {
  Plan& P = *(Plan*)h;
  for (int l = 0; l < P.n_levels; ++l) cnt[l] = P.flevel_ptr[l + 1] - P.flevel_ptr[l];
}

// One instance.  can: canonical values (ncan).  U: usize, Dinv: 3*npiv, S: nc*nc (row-major,
// lower filled; contribution -A K^-1 A^T), inertia[3] += (pos, neg, zero).
int ppsim_factor(void* h, const double* can, double* U, double* Dinv, double* S, int64_t* inertia, double eps) {
  Plan& P = *(Plan*)h;
  std::memset(U, 0, sizeof(double) * P.usize);
  for (int e = 0; e < P.ncan; ++e) U[P.pos_of_can[e]] += can[e];
  std::vector<double> acc;
  int pos = 0, neg = 0, zero = 0;
  for (const auto& t : P.ftasks) {
    const int p = t.piv, w = P.piv_w[p];
    const int64_t off = P.piv_uoff[p];
    const int nr = t.r1 - t.r0;
    acc.assign((size_t)nr * w, 0.0);
    for (int r = 0; r < nr; ++r)
      for (int q = 0; q < w; ++q) acc[r * w + q] = U[off + (int64_t)(t.r0 + r) * w + q];
    double dorig[3] = {0, 0, 0};
    if (t.r0 == 0) { dorig[0] = acc[0]; if (w == 2) { dorig[1] = acc[2]; dorig[2] = acc[3]; } }
    for (int s = t.src0; s < t.src1; ++s) {
      const auto& src = P.fsrcs[s];
      const int k = src.k, wk = P.piv_w[k];
      const int64_t offk = P.piv_uoff[k];
      const double* inv = &Dinv[3 * (size_t)k];
      // M[t][q] = sum_t' inv[t][t'] * U_k[mslot+q][t']
      double M[2][2] = {{0, 0}, {0, 0}};
      for (int q = 0; q < w; ++q) {
        const double* uq = &U[offk + (int64_t)(src.mslot + q) * wk];
        if (wk == 1) M[0][q] = inv[0] * uq[0];
        else {
          M[0][q] = inv[0] * uq[0] + inv[1] * uq[1];
          M[1][q] = inv[1] * uq[0] + inv[2] * uq[1];
        }
      }
      for (int ri = src.run0; ri < src.run1; ++ri) {
        const auto& run = P.runs[ri];
        for (int j = 0; j < run.len; ++j) {
          const double* us = &U[offk + (int64_t)(run.src + j) * wk];
          double* a = &acc[(size_t)(run.dst + j) * w];
          for (int q = 0; q < w; ++q) {
            double v = us[0] * M[0][q];
            if (wk == 2) v += us[1] * M[1][q];
            a[q] -= v;
          }
        }
      }
    }
    if (t.r0 == 0) {
      double a = acc[0], b = (w == 2 ? acc[2] : 0.0), c = (w == 2 ? acc[3] : 0.0);
      double colmax = 0;
      for (int r = w; r < nr; ++r)
        for (int q = 0; q < w; ++q) colmax = std::fmax(colmax, std::fabs(acc[r * w + q]));
      colmax = std::fmax(colmax, std::fabs(dorig[0]));
      if (w == 2) colmax = std::fmax(colmax, std::fmax(std::fabs(dorig[1]), std::fabs(dorig[2])));
      pp::PivotResult pr = pp::invert_pivot(w, a, b, c, colmax, eps);
      Dinv[3 * (size_t)p] = pr.i00; Dinv[3 * (size_t)p + 1] = pr.i10; Dinv[3 * (size_t)p + 2] = pr.i11;
      pos += pr.code & 3; neg += (pr.code >> 2) & 3; zero += (pr.code >> 4) & 3;
    }
    for (int r = 0; r < nr; ++r)
      for (int q = 0; q < w; ++q) U[off + (int64_t)(t.r0 + r) * w + q] = acc[r * w + q];
  }
  inertia[0] += pos; inertia[1] += neg; inertia[2] += zero;
  // Schur tiles
  const int T = P.opt.tile, nc = P.nc;
  for (size_t ti = 0; ti < P.stile_a.size(); ++ti) {
    double accS[8][8] = {};
    for (int r = P.stile_ptr[ti]; r < P.stile_ptr[ti + 1]; ++r) {
      const auto& rec = P.stile_rec[r];
      const int p = rec.piv, w = P.piv_w[p];
      const int64_t off = P.piv_uoff[p];
      const double* inv = &Dinv[3 * (size_t)p];
      for (int i = 0; i < T; ++i) {
        if (rec.slotA[i] < 0) continue;
        const double* ua = &U[off + (int64_t)rec.slotA[i] * w];
        double wa0, wa1 = 0;
        if (w == 1) wa0 = ua[0] * inv[0];
        else { wa0 = ua[0] * inv[0] + ua[1] * inv[1]; wa1 = ua[0] * inv[1] + ua[1] * inv[2]; }
        for (int j = 0; j < T; ++j) {
          if (rec.slotB[j] < 0) continue;
          const double* ub = &U[off + (int64_t)rec.slotB[j] * w];
          accS[i][j] -= wa0 * ub[0] + (w == 2 ? wa1 * ub[1] : 0.0);
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
  for (int k = 0; k < P.n; ++k) W[k] = rhs[P.perm[k]];
  for (int li = 0; li < P.npiv; ++li) {
    const int p = P.lvl_piv[li], w = P.piv_w[p], p0 = P.piv_start[p];
    double y[2] = {W[p0], w == 2 ? W[p0 + 1] : 0.0};
    for (int s = P.sfwd_ptr[p]; s < P.sfwd_ptr[p + 1]; ++s) {
      const int k = P.sfwd_k[s], wk = P.piv_w[k], k0 = P.piv_start[k];
      for (int q = 0; q < w; ++q) {
        const double* u = &U[P.piv_uoff[k] + (int64_t)(P.sfwd_mslot[s] + q) * wk];
        y[q] -= u[0] * W[k0] + (wk == 2 ? u[1] * W[k0 + 1] : 0.0);
      }
    }
    const double* inv = &Dinv[3 * (size_t)p];
    if (w == 1) W[p0] = inv[0] * y[0];
    else { W[p0] = inv[0] * y[0] + inv[1] * y[1]; W[p0 + 1] = inv[1] * y[0] + inv[2] * y[1]; }
  }
  for (int c = 0; c < P.nc; ++c) {
    double s = 0;
    for (int t = P.crow_ptr[c]; t < P.crow_ptr[c + 1]; ++t) {
      const int k = P.crow_k[t], wk = P.piv_w[k], k0 = P.piv_start[k];
      const double* u = &U[P.piv_uoff[k] + (int64_t)P.crow_slot[t] * wk];
      s -= u[0] * W[k0] + (wk == 2 ? u[1] * W[k0 + 1] : 0.0);
    }
    W[P.n + c] = s;
  }
}

// backward: W[0,n) = z, W[n..] = x_c on entry; x (original order) on exit
void ppsim_backward(void* h, const double* U, const double* Dinv, double* W, double* x) {
  Plan& P = *(Plan*)h;
  for (int li = P.npiv - 1; li >= 0; --li) {
    const int p = P.lvl_piv[li], w = P.piv_w[p], p0 = P.piv_start[p];
    double g[2] = {0, 0};
    const int nr = P.piv_rowptr[p + 1] - P.piv_rowptr[p];
    const int* ri = &P.rowidx[P.piv_rowptr[p]];
    const double* u = &U[P.piv_uoff[p] + (int64_t)w * w];
    for (int j = 0; j < nr; ++j)
      for (int q = 0; q < w; ++q) g[q] += u[(int64_t)j * w + q] * W[ri[j]];
    const double* inv = &Dinv[3 * (size_t)p];
    if (w == 1) W[p0] -= inv[0] * g[0];
    else {
      W[p0] -= inv[0] * g[0] + inv[1] * g[1];
      W[p0 + 1] -= inv[1] * g[0] + inv[2] * g[1];
    }
  }
  for (int k = 0; k < P.n; ++k) x[P.perm[k]] = W[k];
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
