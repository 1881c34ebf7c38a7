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

}  // namespace pp
