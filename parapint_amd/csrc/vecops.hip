// Vector kernels of the interior-point step on device-resident vectors (interior_point.py:174-317, 655-758).
#include "common.hpp"

namespace {

// ------------------------------------------------------------------------------------------
// f4: the vector work of the step after the solve, on vectors that stay in HBM.  One pass over a family
// of arrays (primals or slacks with their bounds, bound duals and steps) gives the four scalars the interior-point
// loop needs from it: the fraction-to-the-boundary step lengths of the variable and of its bound duals
// (interior_point.py:655-758) and the complementarity residuals max |(x - l) z_l - mu|, max |(u - x) z_u - mu| over
// the finite bounds (interior_point.py:257-266).  Infinite bounds are skipped, as the reference masks them.
// part[4][gridDim.x]: per-workgroup partial results, combined by k_step_stats_final (deterministic).
// max / min that PROPAGATE NaN (fmax / fmin drop it): a NaN in a residual or a step must reach the caller's test, as
// numpy's max does in the reference (interior_point.py:254-301)
__device__ __forceinline__ double nan_max(double a, double b) { return (a != a || b != b) ? NAN : fmax(a, b); }
__device__ __forceinline__ double nan_min(double a, double b) { return (a != a || b != b) ? NAN : fmin(a, b); }

__global__ __launch_bounds__(256) void k_step_stats(size_t n, const double* __restrict__ x, const double* __restrict__ dx,
                                                    const double* __restrict__ xl, const double* __restrict__ xu,
                                                    const double* __restrict__ zl, const double* __restrict__ dzl,
                                                    const double* __restrict__ zu, const double* __restrict__ dzu, double tau,
                                                    double mu, double* __restrict__ part) {
  __shared__ double red[4][256];
  double ap = 1.0, ad = 1.0, cl = 0.0, cu = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const double xi = x[i], di = dx ? dx[i] : 0.0;
    const double lo = xl ? xl[i] : -INFINITY, hi = xu ? xu[i] : INFINITY;
    if (di != di || xi != xi) ap = NAN;
    if (di < 0.0 && lo > -INFINITY) ap = nan_min(ap, -tau * (xi - lo) / di);
    if (di > 0.0 && hi < INFINITY) ap = nan_min(ap, tau * (hi - xi) / di);
    if (zl) {
      const double z = zl[i], dz = dzl ? dzl[i] : 0.0;
      if (dz != dz || z != z) ad = NAN;
      if (dz < 0.0) ad = nan_min(ad, -tau * z / dz);
      if (lo > -INFINITY) cl = nan_max(cl, fabs((xi - lo) * z - mu));
    }
    if (zu) {
      const double z = zu[i], dz = dzu ? dzu[i] : 0.0;
      if (dz != dz || z != z) ad = NAN;
      if (dz < 0.0) ad = nan_min(ad, -tau * z / dz);
      if (hi < INFINITY) cu = nan_max(cu, fabs((hi - xi) * z - mu));
    }
  }
  red[0][threadIdx.x] = ap; red[1][threadIdx.x] = ad; red[2][threadIdx.x] = cl; red[3][threadIdx.x] = cu;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[0][threadIdx.x] = nan_min(red[0][threadIdx.x], red[0][threadIdx.x + s]);
      red[1][threadIdx.x] = nan_min(red[1][threadIdx.x], red[1][threadIdx.x + s]);
      red[2][threadIdx.x] = nan_max(red[2][threadIdx.x], red[2][threadIdx.x + s]);
      red[3][threadIdx.x] = nan_max(red[3][threadIdx.x], red[3][threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x < 4) part[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = red[threadIdx.x][0];
}

__global__ __launch_bounds__(256) void k_step_stats_final(int nparts, const double* __restrict__ part, double* __restrict__ out) {
  __shared__ double red[4][256];
  double v[4] = {1.0, 1.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < nparts; i += 256) {
    v[0] = nan_min(v[0], part[i]); v[1] = nan_min(v[1], part[(size_t)nparts + i]);
    v[2] = nan_max(v[2], part[2 * (size_t)nparts + i]); v[3] = nan_max(v[3], part[3 * (size_t)nparts + i]);
  }
  for (int q = 0; q < 4; ++q) red[q][threadIdx.x] = v[q];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[0][threadIdx.x] = nan_min(red[0][threadIdx.x], red[0][threadIdx.x + s]);
      red[1][threadIdx.x] = nan_min(red[1][threadIdx.x], red[1][threadIdx.x + s]);
      red[2][threadIdx.x] = nan_max(red[2][threadIdx.x], red[2][threadIdx.x + s]);
      red[3][threadIdx.x] = nan_max(red[3][threadIdx.x], red[3][threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x < 4) out[threadIdx.x] = red[threadIdx.x][0];
}

// y <- y + alpha x  (the primal / dual step, interior_point.py:619-626) and max |v| (infeasibility norms, :268-283)
__global__ __launch_bounds__(256) void k_vec_axpy(size_t n, double alpha, const double* __restrict__ x, double* __restrict__ y) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] += alpha * x[i];
}

__global__ __launch_bounds__(256) void k_vec_max_abs(size_t n, const double* __restrict__ v, double* __restrict__ part) {
  __shared__ double red[256];
  double m = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = nan_max(m, fabs(v[i]));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = nan_max(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {     // slot layout of k_step_stats_final: {min, min, max, max}; the result travels in slot 2
    part[blockIdx.x] = 1.0; part[(size_t)gridDim.x + blockIdx.x] = 1.0;
    part[2 * (size_t)gridDim.x + blockIdx.x] = red[0]; part[3 * (size_t)gridDim.x + blockIdx.x] = 0.0;
  }
}


}  // namespace

// dst[idx[i]] = src[i] resp. dst[i] = src[idx[i]]: the ordering of the coupling variables under which S is block
// tridiagonal (padded) <-> the caller's ordering
__global__ __launch_bounds__(256) void k_vec_scatter(size_t n, const int64_t* __restrict__ idx, const double* __restrict__ src,
                                                     double* __restrict__ dst) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[idx[i]] = src[i];
}
__global__ __launch_bounds__(256) void k_vec_gather(size_t n, const int64_t* __restrict__ idx, const double* __restrict__ src,
                                                    double* __restrict__ dst) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[idx[i]];
}

extern "C" {

// ---- f4: vector kernels of the step after the solve (device-resident vectors) ------------------------------------
static int vec_scratch(pp_handle h, int nblocks, double** part, double** out) {
  if (!h->vec_part) {
    void* p = nullptr;
    if (hipMalloc(&p, (4 * 2048 + 8) * sizeof(double)) != hipSuccess) return fail(h, 1, "hipMalloc failed (vector scratch)");
    h->vec_part = (double*)p;
  }
  (void)nblocks;
  *part = h->vec_part;
  *out = h->vec_part + 4 * 2048;
  return 0;
}

int pp_vec_step_stats(pp_handle h, int64_t n, const double* x, const double* dx, const double* xl, const double* xu,
                      const double* zl, const double* dzl, const double* zu, const double* dzu, double tau, double mu,
                      double out_host[4]) {
  if (!h || n < 0 || !x || !out_host) return fail(h, 3, "pp_vec_step_stats: bad arguments");
  PP_HIP(hipSetDevice(h->device));
  out_host[0] = out_host[1] = 1.0; out_host[2] = out_host[3] = 0.0;
  if (n == 0) return 0;
  const int nb = (int)std::min<int64_t>(2048, (n + 255) / 256);
  double *part, *out;
  if (int rc = vec_scratch(h, nb, &part, &out)) return rc;
  hipLaunchKernelGGL(k_step_stats, dim3(nb), dim3(256), 0, h->stream, (size_t)n, x, dx, xl, xu, zl, dzl, zu, dzu, tau, mu, part);
  hipLaunchKernelGGL(k_step_stats_final, dim3(1), dim3(256), 0, h->stream, nb, part, out);
  PP_HIP(hipGetLastError());
  PP_HIP(hipMemcpyAsync(out_host, out, 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  PP_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

int pp_vec_max_abs(pp_handle h, int64_t n, const double* v, double* out_host) {
  if (!h || n < 0 || !out_host || (n > 0 && !v)) return fail(h, 3, "pp_vec_max_abs: bad arguments");
  PP_HIP(hipSetDevice(h->device));
  *out_host = 0.0;
  if (n == 0) return 0;
  const int nb = (int)std::min<int64_t>(2048, (n + 255) / 256);
  double *part, *out;
  if (int rc = vec_scratch(h, nb, &part, &out)) return rc;
  hipLaunchKernelGGL(k_vec_max_abs, dim3(nb), dim3(256), 0, h->stream, (size_t)n, v, part);
  hipLaunchKernelGGL(k_step_stats_final, dim3(1), dim3(256), 0, h->stream, nb, part, out);
  PP_HIP(hipGetLastError());
  double res[4];
  PP_HIP(hipMemcpyAsync(res, out, 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  PP_HIP(hipStreamSynchronize(h->stream));
  *out_host = res[2];
  return 0;
}

int pp_vec_axpy(pp_handle h, int64_t n, double alpha, const double* x, double* y) {
  if (!h || n < 0 || (n > 0 && (!x || !y))) return fail(h, 3, "pp_vec_axpy: bad arguments");
  PP_HIP(hipSetDevice(h->device));
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_vec_axpy, dim3((unsigned)std::min<int64_t>(4096, (n + 255) / 256)), dim3(256), 0, h->stream, (size_t)n,
                     alpha, x, y);
  PP_HIP(hipGetLastError());
  return 0;
}

int pp_vec_permute(pp_handle h, int64_t n, const int64_t* idx, const double* src, double* dst, int64_t ndst, int scatter) {
  if (!h || n < 0 || ndst < 0 || (n > 0 && (!idx || !src || !dst))) return fail(h, 3, "pp_vec_permute: bad arguments");
  PP_HIP(hipSetDevice(h->device));
  if (scatter && ndst > 0) PP_HIP(hipMemsetAsync(dst, 0, (size_t)ndst * sizeof(double), h->stream));
  if (n == 0) return 0;
  const dim3 grid((unsigned)std::min<int64_t>(1024, (n + 255) / 256));
  if (scatter) hipLaunchKernelGGL(k_vec_scatter, grid, dim3(256), 0, h->stream, (size_t)n, idx, src, dst);
  else hipLaunchKernelGGL(k_vec_gather, grid, dim3(256), 0, h->stream, (size_t)n, idx, src, dst);
  PP_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
