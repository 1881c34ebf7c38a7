// Layout changes between the caller's [instance][entry] arrays and the kernels' [entry][instance] arrays, assembly of
// device-resident sources, diagonal shifts (included by the translation units that launch them).
#pragma once
#include "common.hpp"

namespace {

// ------------------------------------------------------------------------------------------
// [rows][m] row-major  ->  [m'][bpad] (instance-interleaved), zero padding for rows >= nrows.
// rowmap (may be null): entry e of the input goes to output row rowmap[e]; negative = not needed
// (e.g. the upper-triangle half of a KKT block given with both triangles) and is not written.
__global__ __launch_bounds__(256) void k_transpose_in(const double* __restrict__ in, double* __restrict__ out,
                                                      const int* __restrict__ rowmap, int nrows, int m, int bpad,
                                                      int tiles, const int* __restrict__ tile_list) {
  // One workgroup walks `tiles` consecutive 64 x 64 tiles along the entry axis: the rows of the input are
  // read in runs of tiles * 512 bytes, and the loads of the next tile are in flight while the current one
  // goes out through LDS.
  __shared__ double tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int nchunk = bpad / 64;
  const int b0 = PP_CHUNK_OF_WG(nchunk) * 64;
  // tile_list (may be null): only these 64-entry tiles hold entries that are needed (a KKT block given with both
  // triangles has whole runs of upper-triangle entries: those tiles are never read)
  const int tsel = PP_TASK_OF_WG(nchunk);
  int e0 = (tile_list ? tile_list[tsel] : tsel) * 64 * tiles;
  double v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int b = b0 + ty + 4 * q, e = e0 + tx;
    v[q] = (b < nrows && e < m) ? in[(size_t)b * m + e] : 0.0;
  }
  for (int t = 0; t < tiles && e0 < m; ++t, e0 += 64) {
#pragma unroll
    for (int q = 0; q < 16; ++q) tile[ty + 4 * q][tx] = v[q];
    __syncthreads();
    if (t + 1 < tiles) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int b = b0 + ty + 4 * q, e = e0 + 64 + tx;
        v[q] = (b < nrows && e < m) ? in[(size_t)b * m + e] : 0.0;
      }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int r = ty + 4 * q, e = e0 + r, b = b0 + tx;
      if (e < m) {
        const int orow = rowmap ? rowmap[e] : e;
        if (orow >= 0) out[(size_t)orow * bpad + b] = tile[tx][r];
      }
    }
    __syncthreads();
  }
}

// f2 (sc_ip_interface.py:1677-1710, interface.py:432-494): the values of K_i / A_i straight from the
// producer's arrays.  Every used raw entry r is coef[r] * S[src[r]][b] (src < 0: the constant coef[r]); S is
// [source][instance], so reads and writes are coalesced and no transposition is needed.
__global__ __launch_bounds__(256) void k_assemble_sources(const double* __restrict__ S, double* __restrict__ out,
                                                          const int* __restrict__ src, const double* __restrict__ coef,
                                                          int nrows, int bpad) {
  const int nchunk4 = bpad / 64;
  const int r = PP_TASK_OF_WG(nchunk4) * 4 + (threadIdx.x >> 6);
  const int b = PP_CHUNK_OF_WG(nchunk4) * 64 + (threadIdx.x & 63);
  if (r >= nrows) return;
  const int sidx = src[r];
  const double c = coef[r];
  out[(size_t)r * bpad + b] = (sidx >= 0) ? c * S[(size_t)sidx * bpad + b] : c;
}

// Inertia-correction fast path (interior_point.py:364-392, interface.py:590-619): the diagonal entries of the
// rows of class 1 (Hessian) get + delta_w, those of class 2 (constraints) get - delta_c, directly in the
// transposed input of values that are already resident.
__global__ __launch_bounds__(256) void k_shift_diag(double* __restrict__ rawT, const int* __restrict__ rows,
                                                    const int* __restrict__ cls, int nshift, int bpad, double dw,
                                                    double dc) {
  const int j = blockIdx.x, b = blockIdx.y * 256 + threadIdx.x;
  if (j >= nshift || b >= bpad) return;
  rawT[(size_t)rows[j] * bpad + b] += (cls[j] == 1) ? dw : -dc;
}

// out[b][i] = W[iperm[i]][b]
__global__ __launch_bounds__(256) void k_transpose_out(const double* __restrict__ W, const int* __restrict__ iperm,
                                                       double* __restrict__ out, int nrows, int m, int bpad) {
  __shared__ double tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int nchunk = bpad / 64;
  const int i0 = PP_TASK_OF_WG(nchunk) * 64, b0 = PP_CHUNK_OF_WG(nchunk) * 64;
  int src[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) { const int i = i0 + ty + 4 * q; src[q] = (i < m) ? iperm[i] : -1; }
  double v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) v[q] = (src[q] >= 0) ? W[(size_t)src[q] * bpad + b0 + tx] : 0.0;   // all loads in flight
#pragma unroll
  for (int q = 0; q < 16; ++q) tile[ty + 4 * q][tx] = v[q];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int r = ty + 4 * q, b = b0 + r, i = i0 + tx;
    if (b < nrows && i < m) out[(size_t)b * m + i] = tile[tx][r];
  }
}


}  // namespace
