// Static-pivot inversion rule shared by the HIP kernels and the test-only host interpreter.
//
// Role of MA27B's pivot test + info(15) (reference: parapint/linalg/ma27_interface.py:124-136,
// 201-203): a 1x1 / 2x2 pivot block is inverted; a numerically zero pivot is replaced by a tiny
// one and counted as a zero eigenvalue so the caller reports LinearSolverStatus.singular and the
// interior-point inertia-correction loop regularises (interior_point.py:364-399).
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define PP_HD __host__ __device__ __forceinline__
#else
#define PP_HD inline
#endif

namespace pp {

// inertia contribution encoded as pos + 4*neg + 16*zero (each 0..2)
struct PivotResult { double i00, i10, i11; int code; };

PP_HD PivotResult invert_pivot(int w, double a, double b, double c, double colmax, double eps) {
  PivotResult r;
  if (w == 1) {
    double ref = fmax(fabs(a), colmax);
    if (!(fabs(a) > eps * ref) || ref == 0.0) {
      a = (ref > 0.0 ? eps * ref : 1.0);
      r.code = 16;
    } else {
      r.code = (a > 0.0) ? 1 : 4;
    }
    r.i00 = 1.0 / a; r.i10 = 0.0; r.i11 = 0.0;
  } else {
    double det = a * c - b * b;
    double ref = fmax(fabs(a * c), b * b);
    if (!(fabs(det) > eps * ref) || ref == 0.0) {
      r.code = 32;
      det = (ref > 0.0 ? eps * ref : 1.0);
      if (ref == 0.0) { a = 1.0; c = 1.0; b = 0.0; }
    } else if (det < 0.0) {
      r.code = 1 + 4;
    } else {
      r.code = (a > 0.0) ? 2 : 8;
    }
    r.i00 = c / det; r.i10 = -b / det; r.i11 = a / det;
  }
  return r;
}

// ---------------------------------------------------------------------------------------------
// Block pivot (supernode) of width w <= WB (template bound, <= PP_WMAX) made of a static sequence of 1x1 / 2x2 sub-pivots
// (bit i of `sub` set: columns i, i+1 form a 2x2 sub-pivot).  The symmetric sweep operator is
// applied sub-pivot by sub-pivot; the values met on the diagonal are exactly the LDL^T pivots of
// the static sequence, so inertia and the zero-pivot rule are those of invert_pivot.  On exit
// `inv` holds inv(A) packed by rows of the lower triangle (t, t' <= t at t(t+1)/2 + t').
// code: pos | neg << 4 | zero << 8.
// tmd[i]: the largest magnitude among the terms that were summed into diagonal entry (i, i) when the block was
// gathered.  The zero test of a 1x1 sub-pivot compares it with the largest term of ITS OWN sum -- the gathered ones and
// the updates it received inside the block -- as the gather kernels do for a scalar pivot: a pivot is numerically
// zero when it is rounding noise of what was added up to form it.  (Rounds 1-2 took one bound for the whole block; a
// block pivot that holds a primal variable with a barrier weight of 1e7 next to a constraint row with -1e-7 then
// reported the second, perfectly accurate, pivot as zero: round 3)
#ifndef PP_WMAX
#define PP_WMAX 4
#endif

template <int WB>
PP_HD int invert_block_t(int w, unsigned sub, const double* a /* row-major with stride WB, lower triangle read */,
                         const double* tmd /* [WB] */, double eps, double* inv) {
  double A[WB][WB], m[WB];
#pragma unroll
  for (int i = 0; i < WB; ++i) m[i] = (i < w) ? tmd[i] : 0.0;
#pragma unroll
  for (int i = 0; i < WB; ++i)
#pragma unroll
    for (int j = 0; j < WB; ++j) {
      const int hi = i > j ? i : j, lo = i > j ? j : i;
      A[i][j] = (hi < w) ? a[hi * WB + lo] : ((i == j) ? 1.0 : 0.0);
    }
  int pos = 0, neg = 0, zero = 0;
  bool second = false;   // current column is the second column of a 2x2 sub-pivot
#pragma unroll
  for (int k = 0; k < WB; ++k) {
    if (k >= w) continue;
    if (second) { second = false; continue; }
    const bool two = ((sub >> k) & 1u) && (k + 1 < w);
    if (!two) {
      const PivotResult pr = invert_pivot(1, A[k][k], 0.0, 0.0, m[k], eps);
      pos += pr.code & 3; neg += (pr.code >> 2) & 3; zero += (pr.code >> 4) & 3;
      double l[WB];
#pragma unroll
      for (int i = 0; i < WB; ++i) l[i] = A[i][k] * pr.i00;
#pragma unroll
      for (int i = 0; i < WB; ++i)
        if (i != k) m[i] = fmax(m[i], fabs(l[i] * A[k][i]));
#pragma unroll
      for (int i = 0; i < WB; ++i)
#pragma unroll
        for (int j = 0; j < WB; ++j)
          if (i != k && j != k) A[i][j] -= l[i] * A[k][j];
#pragma unroll
      for (int i = 0; i < WB; ++i)
        if (i != k) { A[i][k] = l[i]; A[k][i] = l[i]; }
      A[k][k] = -pr.i00;
    } else {
      second = true;
      const int k1 = k + 1 < WB ? k + 1 : k;
      const PivotResult pr = invert_pivot(2, A[k][k], A[k1][k], A[k1][k1], 0.0, eps);   // (a 2x2 pivot is tested by its determinant)
      pos += pr.code & 3; neg += (pr.code >> 2) & 3; zero += (pr.code >> 4) & 3;
      double l0[WB], l1[WB];
#pragma unroll
      for (int i = 0; i < WB; ++i) {
        l0[i] = A[i][k] * pr.i00 + A[i][k1] * pr.i10;
        l1[i] = A[i][k] * pr.i10 + A[i][k1] * pr.i11;
      }
#pragma unroll
      for (int i = 0; i < WB; ++i)
        if (i != k && i != k1) m[i] = fmax(m[i], fmax(fabs(l0[i] * A[k][i]), fabs(l1[i] * A[k1][i])));
#pragma unroll
      for (int i = 0; i < WB; ++i)
#pragma unroll
        for (int j = 0; j < WB; ++j)
          if (i != k && i != k1 && j != k && j != k1) A[i][j] -= l0[i] * A[k][j] + l1[i] * A[k1][j];
#pragma unroll
      for (int i = 0; i < WB; ++i)
        if (i != k && i != k1) { A[i][k] = l0[i]; A[k][i] = l0[i]; A[i][k1] = l1[i]; A[k1][i] = l1[i]; }
      A[k][k] = -pr.i00; A[k1][k] = -pr.i10; A[k][k1] = -pr.i10; A[k1][k1] = -pr.i11;
    }
  }
#pragma unroll
  for (int i = 0; i < WB; ++i)
#pragma unroll
    for (int j = 0; j < WB; ++j)
      if (j <= i && i < w) inv[i * (i + 1) / 2 + j] = -A[i][j];
  return pos | (neg << 4) | (zero << 8);
}

// ---------------------------------------------------------------------------------------------
// Root front (plan.hpp, front_piv): a block pivot of up to WF = 16 columns, the same static sequence of 1x1 / 2x2
// sub-pivots and the same sweep operator, written the way the device runs it (k_front_invert: one wave per row of A,
// the pivot rows broadcast through LDS): row i is updated from its own column entries and the OLD pivot row(s); the
// pivot rows themselves are scaled in place.  A is the full symmetric matrix, row-major with stride WF; inv is
// packed like invert_block_t's.  This function is the definition the kernel is tested against (tests/hostsim).
constexpr int PP_WF = 16;
PP_HD int invert_front(int w, unsigned sub, double* A /* [PP_WF * PP_WF], rows >= w ignored */, const double* tmd /* [PP_WF] */,
                       double eps, double* inv) {
  int pos = 0, neg = 0, zero = 0;
  double m[PP_WF];
  for (int i = 0; i < PP_WF; ++i) m[i] = (i < w) ? tmd[i] : 0.0;
  for (int k = 0; k < w; ++k) {
    const bool two = ((sub >> k) & 1u) && (k + 1 < w);
    if (!two) {
      const PivotResult pr = invert_pivot(1, A[k * PP_WF + k], 0.0, 0.0, m[k], eps);
      pos += pr.code & 3; neg += (pr.code >> 2) & 3; zero += (pr.code >> 4) & 3;
      for (int i = 0; i < w; ++i) {
        if (i == k) continue;
        const double l = A[i * PP_WF + k] * pr.i00;
        m[i] = fmax(m[i], fabs(l * A[k * PP_WF + i]));
        for (int j = 0; j < w; ++j)
          if (j != k) A[i * PP_WF + j] -= l * A[k * PP_WF + j];
        A[i * PP_WF + k] = l;
      }
      for (int j = 0; j < w; ++j)
        if (j != k) A[k * PP_WF + j] = A[k * PP_WF + j] * pr.i00;
      A[k * PP_WF + k] = -pr.i00;
    } else {
      const int k1 = k + 1;
      const PivotResult pr = invert_pivot(2, A[k * PP_WF + k], A[k1 * PP_WF + k], A[k1 * PP_WF + k1], 0.0, eps);
      pos += pr.code & 3; neg += (pr.code >> 2) & 3; zero += (pr.code >> 4) & 3;
      for (int i = 0; i < w; ++i) {
        if (i == k || i == k1) continue;
        const double l0 = A[i * PP_WF + k] * pr.i00 + A[i * PP_WF + k1] * pr.i10;
        const double l1 = A[i * PP_WF + k] * pr.i10 + A[i * PP_WF + k1] * pr.i11;
        m[i] = fmax(m[i], fmax(fabs(l0 * A[k * PP_WF + i]), fabs(l1 * A[k1 * PP_WF + i])));
        for (int j = 0; j < w; ++j)
          if (j != k && j != k1) A[i * PP_WF + j] -= l0 * A[k * PP_WF + j] + l1 * A[k1 * PP_WF + j];
        A[i * PP_WF + k] = l0; A[i * PP_WF + k1] = l1;
      }
      for (int j = 0; j < w; ++j) {
        if (j == k || j == k1) continue;
        const double a0 = A[k * PP_WF + j], a1 = A[k1 * PP_WF + j];
        A[k * PP_WF + j] = a0 * pr.i00 + a1 * pr.i10;
        A[k1 * PP_WF + j] = a0 * pr.i10 + a1 * pr.i11;
      }
      A[k * PP_WF + k] = -pr.i00; A[k1 * PP_WF + k] = -pr.i10; A[k * PP_WF + k1] = -pr.i10; A[k1 * PP_WF + k1] = -pr.i11;
      ++k;
    }
  }
  for (int i = 0; i < w; ++i)
    for (int j = 0; j <= i; ++j) inv[i * (i + 1) / 2 + j] = -A[i * PP_WF + j];
  return pos | (neg << 4) | (zero << 8);
}

// bound = the widest block the build supports
PP_HD int invert_block(int w, unsigned sub, const double* a /* stride PP_WMAX */, const double* tmd /* [PP_WMAX] */, double eps,
                       double* inv) {
  return invert_block_t<PP_WMAX>(w, sub, a, tmd, eps, inv);
}

}  // namespace pp
