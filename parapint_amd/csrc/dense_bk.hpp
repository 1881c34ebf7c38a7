// Dense symmetric-indefinite LDL^T (Bunch-Kaufman partial pivoting, lower storage) and the
// matching solve, for the assembled Schur complement S (n_c x n_c, replicated on every rank).
//
// Replaces the reference's `schur_complement_solver` sub-solver on S
// (mpi_explicit_schur_complement.py:352-361, 391; MA27 / SuperLU on a COO S) and supplies
// inertia(S) for get_inertia (mpi_...:431-434).  The algorithm is the classical unblocked
// Bunch-Kaufman (alpha = (1+sqrt(17))/8) with 1x1 and 2x2 pivots; it is written once against a
// tiny "thread team" context so that the same source runs single-threaded in the CPU tests and
// as one 1024-thread workgroup on the GPU.
//
// Ctx must provide: tid(), nthreads(), sync(), argmax(double v, int i, double* vmax, int* imax),
// maxval(double v) -> double, sum(double v) -> double (team-wide, result known to all threads).
#pragma once
#include "pivot.hpp"

namespace pp {

struct BkInfo { int npos, nneg, nzero; };

template <class Ctx>
PP_HD void bk_factor(Ctx& ctx, int n, double* A, int lda, int* ipiv, double* work /*2n*/, BkInfo* info,
                     double eps) {
  const double alpha = 0.6403882032022076;  // (1 + sqrt(17)) / 8
  const int tid = ctx.tid(), nt = ctx.nthreads();
  const int lanes = nt < 64 ? nt : 64;
  const int lane = tid % lanes, wv = tid / lanes, nwv = nt / lanes;
  int npos = 0, nneg = 0, nzero = 0;
  // overall scale for the singularity test
  double loc = 0.0;
  for (int j = 0; j < n; ++j)
    for (int i = j + tid; i < n; i += nt) loc = fmax(loc, fabs(A[i + (size_t)j * lda]));
  const double anorm = ctx.maxval(loc);
  int k = 0;
  while (k < n) {
    int kstep = 1, kp = k;
    const double absakk = fabs(A[k + (size_t)k * lda]);
    double lm = -1.0; int li = k;
    for (int i = k + 1 + tid; i < n; i += nt) {
      double v = fabs(A[i + (size_t)k * lda]);
      if (v > lm) { lm = v; li = i; }
    }
    double colmax; int imax;
    ctx.argmax(lm, li, &colmax, &imax);
    if (colmax < 0.0) colmax = 0.0;
    bool zero_pivot = false;
    if (!(fmax(absakk, colmax) > eps * anorm) || anorm == 0.0) {
      zero_pivot = true;   // column is numerically zero: singular S
    } else {
      if (absakk >= alpha * colmax) {
        kp = k;
      } else {
        double lr = 0.0;
        for (int j = k + tid; j < imax; j += nt) lr = fmax(lr, fabs(A[imax + (size_t)j * lda]));
        for (int i = imax + 1 + tid; i < n; i += nt) lr = fmax(lr, fabs(A[i + (size_t)imax * lda]));
        const double rowmax = ctx.maxval(lr);
        if (absakk >= alpha * colmax * (colmax / rowmax)) kp = k;
        else if (fabs(A[imax + (size_t)imax * lda]) >= alpha * rowmax) kp = imax;
        else { kp = imax; kstep = 2; }
      }
      const int kk = k + kstep - 1;
      if (kp != kk) {
        // symmetric interchange of rows/columns kk and kp inside A(k:n, k:n)
        for (int i = kp + 1 + tid; i < n; i += nt) {
          double t = A[i + (size_t)kk * lda]; A[i + (size_t)kk * lda] = A[i + (size_t)kp * lda];
          A[i + (size_t)kp * lda] = t;
        }
        for (int j = kk + 1 + tid; j < kp; j += nt) {
          double t = A[j + (size_t)kk * lda]; A[j + (size_t)kk * lda] = A[kp + (size_t)j * lda];
          A[kp + (size_t)j * lda] = t;
        }
        if (tid == 0) {
          double t = A[kk + (size_t)kk * lda]; A[kk + (size_t)kk * lda] = A[kp + (size_t)kp * lda];
          A[kp + (size_t)kp * lda] = t;
          if (kstep == 2) {
            t = A[k + 1 + (size_t)k * lda]; A[k + 1 + (size_t)k * lda] = A[kp + (size_t)k * lda];
            A[kp + (size_t)k * lda] = t;
          }
        }
        ctx.sync();
      }
    }
    if (zero_pivot) {
      // leave a unit pivot so the solve stays finite; counted as a zero eigenvalue
      if (tid == 0) A[k + (size_t)k * lda] = (anorm > 0.0 ? anorm : 1.0);
      for (int i = k + 1 + tid; i < n; i += nt) A[i + (size_t)k * lda] = 0.0;
      nzero += 1;
      if (tid == 0) ipiv[k] = k;
      ctx.sync();
      k += 1;
      continue;
    }
    if (kstep == 1) {
      const double akk = A[k + (size_t)k * lda];
      if (akk > 0.0) npos++; else nneg++;
      const double r1 = 1.0 / akk;
      // trailing update with the unscaled column, then scale the column
      for (int j = k + 1 + wv; j < n; j += nwv) {
        const double aj = A[j + (size_t)k * lda] * r1;
        for (int i = j + lane; i < n; i += lanes) A[i + (size_t)j * lda] -= A[i + (size_t)k * lda] * aj;
      }
      ctx.sync();
      for (int i = k + 1 + tid; i < n; i += nt) A[i + (size_t)k * lda] *= r1;
      if (tid == 0) ipiv[k] = kp;
    } else {
      const double a11 = A[k + (size_t)k * lda], a21 = A[k + 1 + (size_t)k * lda];
      const double a22 = A[k + 1 + (size_t)(k + 1) * lda];
      const double det = a11 * a22 - a21 * a21;
      if (det < 0.0) { npos++; nneg++; } else if (a11 > 0.0) npos += 2; else nneg += 2;
      double d21 = a21;
      const double d11 = a22 / d21, d22 = a11 / d21;
      const double t = 1.0 / (d11 * d22 - 1.0);
      d21 = t / d21;
      for (int j = k + 2 + tid; j < n; j += nt) {
        const double ajk = A[j + (size_t)k * lda], ajk1 = A[j + (size_t)(k + 1) * lda];
        work[j] = d21 * (d11 * ajk - ajk1);
        work[n + j] = d21 * (d22 * ajk1 - ajk);
      }
      ctx.sync();
      for (int j = k + 2 + wv; j < n; j += nwv) {
        const double wk = work[j], wk1 = work[n + j];
        for (int i = j + lane; i < n; i += lanes)
          A[i + (size_t)j * lda] -= A[i + (size_t)k * lda] * wk + A[i + (size_t)(k + 1) * lda] * wk1;
      }
      ctx.sync();
      for (int j = k + 2 + tid; j < n; j += nt) {
        A[j + (size_t)k * lda] = work[j];
        A[j + (size_t)(k + 1) * lda] = work[n + j];
      }
      if (tid == 0) { ipiv[k] = -(kp + 1); ipiv[k + 1] = -(kp + 1); }
    }
    ctx.sync();
    k += kstep;
  }
  if (tid == 0) { info->npos = npos; info->nneg = nneg; info->nzero = nzero; }
  ctx.sync();
}

// Solve S x = b in place (b length n) with the factor from bk_factor.
template <class Ctx>
PP_HD void bk_solve(Ctx& ctx, int n, const double* A, int lda, const int* ipiv, double* b) {
  const int tid = ctx.tid(), nt = ctx.nthreads();
  // forward: L D y = P^T b
  int k = 0;
  while (k < n) {
    if (ipiv[k] >= 0) {
      const int kp = ipiv[k];
      if (kp != k && tid == 0) { double t = b[k]; b[k] = b[kp]; b[kp] = t; }
      ctx.sync();
      const double bk = b[k];
      for (int i = k + 1 + tid; i < n; i += nt) b[i] -= A[i + (size_t)k * lda] * bk;
      ctx.sync();
      if (tid == 0) b[k] = bk / A[k + (size_t)k * lda];
      k += 1;
    } else {
      const int kp = -ipiv[k] - 1;
      if (kp != k + 1 && tid == 0) { double t = b[k + 1]; b[k + 1] = b[kp]; b[kp] = t; }
      ctx.sync();
      const double bk = b[k], bk1 = b[k + 1];
      for (int i = k + 2 + tid; i < n; i += nt)
        b[i] -= A[i + (size_t)k * lda] * bk + A[i + (size_t)(k + 1) * lda] * bk1;
      ctx.sync();
      if (tid == 0) {
        const double akm1k = A[k + 1 + (size_t)k * lda];
        const double akm1 = A[k + (size_t)k * lda] / akm1k, ak = A[k + 1 + (size_t)(k + 1) * lda] / akm1k;
        const double denom = akm1 * ak - 1.0;
        const double bkm1 = bk / akm1k, bkk = bk1 / akm1k;
        b[k] = (ak * bkm1 - bkk) / denom;
        b[k + 1] = (akm1 * bkk - bkm1) / denom;
      }
      k += 2;
    }
    ctx.sync();
  }
  // backward: L^T x = y, then undo the interchanges
  k = n - 1;
  while (k >= 0) {
    if (ipiv[k] >= 0) {
      double s = 0.0;
      for (int i = k + 1 + tid; i < n; i += nt) s += A[i + (size_t)k * lda] * b[i];
      s = ctx.sum(s);
      if (tid == 0) {
        b[k] -= s;
        const int kp = ipiv[k];
        if (kp != k) { double t = b[k]; b[k] = b[kp]; b[kp] = t; }
      }
      k -= 1;
    } else {
      double s0 = 0.0, s1 = 0.0;
      for (int i = k + 1 + tid; i < n; i += nt) {
        s0 += A[i + (size_t)k * lda] * b[i];
        s1 += A[i + (size_t)(k - 1) * lda] * b[i];
      }
      s0 = ctx.sum(s0);
      s1 = ctx.sum(s1);
      if (tid == 0) {
        b[k] -= s0;
        b[k - 1] -= s1;
        const int kp = -ipiv[k] - 1;
        if (kp != k) { double t = b[k]; b[k] = b[kp]; b[kp] = t; }
      }
      k -= 2;
    }
    ctx.sync();
  }
}

}  // namespace pp
