// A caller's DEVICE MODEL written in HIP: the functions of the Burgers control problem (parapint/examples/burgers.py:63-176,
// restated in parapint_amd/examples/burgers.py) for all time blocks of a pattern group at once, on the [row][lane] arrays of the
// interior-point step (lane = time block; include/parapint_hip.h: pp_ip_group with obj_row >= 0).  What Pyomo / ASL evaluate for
// the reference at every iterate -- grad f, c(x), the Jacobian and the Hessian of the Lagrangian -- is here one pass over the
// iterate that never leaves the device.  The model is the caller's code, not part of the solver: it lives in the library only
// so that the example has no build step of its own.
//   W     x = y[k][i] (k = 0 .. nt, i = 0 .. m - 1), then u[k][i]; the equality multipliers from row y_eq on
//   data  grad f (n rows) | -c(x) (nt m rows, then the 2 m rows of the initial conditions) | ... | objective per lane (obj_row)
//   src   Hessian values from row hess (the n diagonal entries are constant; then nt (m - 1) entries of the convective term),
//         Jacobian values from row jac in the entry order of BurgersNLP._patterns (5 m - 2 per time node)
#include "common.hpp"

#pragma clang fp contract(off)       // (the same operations in the same order as the numpy / torch form of the model)

namespace {

struct BurgersArgs {
  int m, nt, n, bpad, y_eq, hess, jac, obj_row, init, start_term;
  double dx, omega, v, r;
  const double* dt;      // [bpad]: the time step of every lane, as its NLP object computed it
  const double* w;       // [nt + 1][bpad]: trapezoid weights in time
  const double* y0;      // [m]
};

// one workgroup = one time node k x 64 lanes; the waves share the grid points
__global__ __launch_bounds__(256) void k_burgers_model(BurgersArgs a, const double* __restrict__ W, double* __restrict__ src,
                                                       double* __restrict__ data, double* __restrict__ part) {
  __shared__ double red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = blockIdx.x % (a.nt + 1);
  const size_t b = (size_t)(blockIdx.x / (a.nt + 1)) * 64 + lane, bp = (size_t)a.bpad;
  const int m = a.m, nt = a.nt;
  const size_t U0 = (size_t)(nt + 1) * m;                      // first row of u
  const double dt = a.dt[b], wk = a.w[(size_t)k * bp + b];
  const double* lam = W + (size_t)a.y_eq * bp;
  double f = 0.0;
  for (int i = wave; i < m; i += 4) {
    const double y = W[((size_t)k * m + i) * bp + b], u = W[(U0 + (size_t)k * m + i) * bp + b];
    const double yr = y - a.y0[i];
    data[((size_t)k * m + i) * bp + b] = a.dx * wk * yr;                                     // grad f, state
    double gu = a.dx * wk * a.omega * u;
    if (a.start_term && k == 0) gu = gu + 0.5 * a.dx * dt * a.omega * u;
    data[(U0 + (size_t)k * m + i) * bp + b] = gu;                                            // grad f, control
    f += wk * (yr * yr + a.omega * (u * u));
    if (k >= 1) {
      const double up = (i + 1 < m) ? W[((size_t)k * m + i + 1) * bp + b] : 0.0;
      const double dn = (i >= 1) ? W[((size_t)k * m + i - 1) * bp + b] : 0.0;
      const double yp = W[((size_t)(k - 1) * m + i) * bp + b], upv = W[(U0 + (size_t)(k - 1) * m + i) * bp + b];
      const double conv = (up - dn) / (2.0 * a.dx);
      const double c = (y - yp) / dt - a.v * (up - 2.0 * y + dn) / (a.dx * a.dx) + conv * y - a.r - upv;
      data[((size_t)a.n + (size_t)(k - 1) * m + i) * bp + b] = -c;
      const size_t J = ((size_t)a.jac + (size_t)(k - 1) * (5 * m - 2)) * bp + b;
      src[J + (size_t)i * bp] = 1.0 / dt + 2.0 * a.v / (a.dx * a.dx) + conv;                 // d / d y[k][i]
      src[J + (size_t)(m + i) * bp] = -1.0 / dt;                                             // d / d y[k-1][i]
      if (i + 1 < m) src[J + (size_t)(2 * m + i) * bp] = -a.v / (a.dx * a.dx) + y / (2.0 * a.dx);        // d / d y[k][i+1]
      if (i >= 1) src[J + (size_t)(3 * m - 1 + i - 1) * bp] = -a.v / (a.dx * a.dx) - y / (2.0 * a.dx);   // d / d y[k][i-1]
      src[J + (size_t)(4 * m - 2 + i) * bp] = -1.0;                                          // d / d u[k-1][i]
      if (i + 1 < m) {
        const double l0 = lam[((size_t)(k - 1) * m + i) * bp + b], l1 = lam[((size_t)(k - 1) * m + i + 1) * bp + b];
        src[((size_t)a.hess + a.n + (size_t)(k - 1) * (m - 1) + i) * bp + b] = (l0 - l1) / (2.0 * a.dx);
      }
    } else if (a.init) {
      data[((size_t)a.n + (size_t)nt * m + i) * bp + b] = -yr;                               // y[0][i] = y0(x_i)
      data[((size_t)a.n + (size_t)nt * m + m + i) * bp + b] = -u;                            // u[0][i] = 0
    }
  }
  // objective: this time node's share per lane (the waves in a fixed order), summed over the nodes by k_burgers_objective
  red[wave][lane] = f;
  __syncthreads();
  if (wave == 0) part[(size_t)k * bp + b] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

__global__ __launch_bounds__(64) void k_burgers_objective(BurgersArgs a, const double* __restrict__ W, const double* __restrict__ part,
                                                          double* __restrict__ data) {
  const size_t b = (size_t)blockIdx.x * 64 + threadIdx.x, bp = (size_t)a.bpad;
  double f = 0.0;
  for (int k = 0; k <= a.nt; ++k) f += part[(size_t)k * bp + b];
  f = 0.5 * a.dx * f;
  if (a.start_term) {
    double s = 0.0;
    const size_t U0 = (size_t)(a.nt + 1) * a.m;
    for (int i = 0; i < a.m; ++i) { const double u = W[(U0 + i) * bp + b]; s += u * u; }
    f = f + 0.25 * a.dx * a.dt[b] * a.omega * s;
  }
  data[(size_t)a.obj_row * bp + b] = f;
}

}  // namespace

extern "C" int pp_example_burgers_model(void* stream, int m, int nt, int n, int bpad, int y_eq, int hess, int jac, int obj_row,
                                        int init_conditions, int start_term, double dx, double omega, double v, double r,
                                        const double* dt_lane, const double* w, const double* y0, const double* W, double* src,
                                        double* data, double* scratch) {
  if (m < 1 || nt < 1 || bpad < 64 || (bpad & 63) || n != 2 * (nt + 1) * m || !dt_lane || !w || !y0 || !W || !src || !data || !scratch)
    return 3;
  BurgersArgs a{m, nt, n, bpad, y_eq, hess, jac, obj_row, init_conditions, start_term, dx, omega, v, r, dt_lane, w, y0};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_burgers_model, dim3((unsigned)(nt + 1) * (unsigned)(bpad / 64)), dim3(256), 0, st, a, W, src, data, scratch);
  hipLaunchKernelGGL(k_burgers_objective, dim3((unsigned)(bpad / 64)), dim3(64), 0, st, a, W, (const double*)scratch, data);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
