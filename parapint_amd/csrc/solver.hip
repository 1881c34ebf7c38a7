// MI355X (gfx950) kernels and C-ABI of the batched Schur-complement KKT solver.
//
// Mapping (see plan.hpp): lane = scenario block.  Every value array is [entry][instance], so
// a wavefront touches 64 consecutive doubles (512 B) per access and all control flow / index
// data is wave-uniform (scalar loads, scalar branches): the sparse phase is a pure HBM/L2
// streaming workload with no divergence.  One 64-thread workgroup = one task x 64 instances.
//
// Kernels (reference function each one replaces):
//   k_transpose_in / k_assemble   values of K_i, A_i -> panel storage   (MA27B input, ma27_interface.py:124)
//   k_factor_level                left-looking LDL^T panel updates, static 1x1/2x2 pivots (MA27B)
//   k_count_codes                 inertia / zero-pivot counts           (ma27_interface.py:201-203)
//   k_schur_tiles, k_schur_reduce S_local = -sum_i A_i K_i^-1 A_i^T     (mpi_explicit_schur_complement.py:312-333)
//   k_bk_factor                   dense LDL^T of S + Q                  (mpi_...:347-361)
//   k_fwd_level, k_fwd_coupling   forward substitution, r_s             (mpi_...:381-385; MA27C)
//   k_coupling_solve              x_c = S^-1 (r_c + r_s)                (mpi_...:388-391)
//   k_bwd_level                   back substitution with x_c            (mpi_...:393-396)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/parapint_hip.h"
#include "dense_bk.hpp"
#include "plan.hpp"

namespace {

constexpr int WAVE = 64;
constexpr int PP_NPHASE = 8;  // assemble, factor, schur, dense, fwd, fwd_coupling, coupling_solve, bwd
constexpr int BK_THREADS = 512;
constexpr double PIVOT_EPS = 1e-13;
constexpr double BK_EPS = 1e-14;

// ------------------------------------------------------------------------------------------
// device image of one group's plan (all pointers are device memory)
struct GroupDev {
  int n, nc, batch, bpad, nchunk, npiv, nraw;
  int64_t usize;
  const int *piv_w, *piv_start, *piv_uoff, *piv_rowptr, *rowidx, *perm, *iperm;
  const int *ftask, *fsrc, *runs;
  const int *lvl_piv, *sfwd_ptr, *sfwd_k, *sfwd_mslot;
  const int *crow_ptr, *crow_k, *crow_slot;
  const int *stile_a, *stile_b, *stile_ptr, *stile_rec;
  const int *asm_ptr, *asm_idx;
  double *raw, *rawT, *U, *Dinv, *W, *rhs, *rhsT, *xout, *Spart, *rspart;
  unsigned char* codes;
};

// ------------------------------------------------------------------------------------------
// [rows][m] row-major  ->  [m][bpad] (instance-interleaved), zero padding for rows >= nrows
__global__ __launch_bounds__(256) void k_transpose_in(const double* __restrict__ in, double* __restrict__ out,
                                                      int nrows, int m, int bpad) {
  __shared__ double tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int e0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
  for (int r = ty; r < 64; r += 4) {
    const int b = b0 + r, e = e0 + tx;
    tile[r][tx] = (b < nrows && e < m) ? in[(size_t)b * m + e] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int e = e0 + r, b = b0 + tx;
    if (e < m) out[(size_t)e * bpad + b] = tile[tx][r];
  }
}

// out[b][i] = W[iperm[i]][b]
__global__ __launch_bounds__(256) void k_transpose_out(const double* __restrict__ W, const int* __restrict__ iperm,
                                                       double* __restrict__ out, int nrows, int m, int bpad) {
  __shared__ double tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int i0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
  for (int r = ty; r < 64; r += 4) {
    const int i = i0 + r;
    tile[r][tx] = (i < m) ? W[(size_t)iperm[i] * bpad + b0 + tx] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int b = b0 + r, i = i0 + tx;
    if (b < nrows && i < m) out[(size_t)b * m + i] = tile[tx][r];
  }
}

// U[pos][b] = sum of the raw entries mapped to pos (0 for fill positions)
__global__ __launch_bounds__(256) void k_assemble(GroupDev g) {
  const int lane = threadIdx.x & 63;
  const int64_t pos = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pos >= g.usize) return;
  const int b = blockIdx.y * 64 + lane;
  double v = 0.0;
  for (int t = g.asm_ptr[pos]; t < g.asm_ptr[pos + 1]; ++t) v += g.rawT[(size_t)g.asm_idx[t] * g.bpad + b];
  g.U[(size_t)pos * g.bpad + b] = v;
}

// ------------------------------------------------------------------------------------------
// One factor task: rows [r0, r1) of panel p gathered from descendant panels (left-looking),
// accumulators in LDS as [slot][lane]; the chunk holding the pivot block inverts it.
__global__ __launch_bounds__(64) void k_factor_level(GroupDev g, int task0, double eps) {
  extern __shared__ __attribute__((aligned(16))) double acc[];
  const int lane = threadIdx.x;
  const int b = blockIdx.y * 64 + lane;
  const size_t bpad = (size_t)g.bpad;
  const int* t = g.ftask + 5 * (size_t)(task0 + blockIdx.x);
  const int p = t[0], r0 = t[1], r1 = t[2], s0 = t[3], s1 = t[4];
  const int w = g.piv_w[p];
  double* Up = g.U + (size_t)g.piv_uoff[p] * bpad + b;
  const int nr = r1 - r0, nacc = nr * w;
  for (int i = 0; i < nacc; ++i) acc[i * 64 + lane] = Up[(size_t)(r0 * w + i) * bpad];
  double d0 = 0.0, d1 = 0.0, d2 = 0.0;
  if (r0 == 0) {
    d0 = acc[lane];
    if (w == 2) { d1 = acc[2 * 64 + lane]; d2 = acc[3 * 64 + lane]; }
  }
  for (int s = s0; s < s1; ++s) {
    const int* src = g.fsrc + 4 * (size_t)s;
    const int k = src[0], mslot = src[1], run0 = src[2], run1 = src[3];
    const int wk = g.piv_w[k];
    const double* Uk = g.U + (size_t)g.piv_uoff[k] * bpad + b;
    const double* inv = g.Dinv + (size_t)3 * k * bpad + b;
    const double i00 = inv[0];
    if (wk == 1) {
      const double m00 = i00 * Uk[(size_t)mslot * bpad];
      if (w == 1) {
        for (int ri = run0; ri < run1; ++ri) {
          const int* run = g.runs + 3 * (size_t)ri;
          const int rs = run[0], rd = run[1], len = run[2];
          for (int j = 0; j < len; ++j) acc[(rd + j) * 64 + lane] -= Uk[(size_t)(rs + j) * bpad] * m00;
        }
      } else {
        const double m01 = i00 * Uk[(size_t)(mslot + 1) * bpad];
        for (int ri = run0; ri < run1; ++ri) {
          const int* run = g.runs + 3 * (size_t)ri;
          const int rs = run[0], rd = run[1], len = run[2];
          for (int j = 0; j < len; ++j) {
            const double us = Uk[(size_t)(rs + j) * bpad];
            acc[((rd + j) * 2) * 64 + lane] -= us * m00;
            acc[((rd + j) * 2 + 1) * 64 + lane] -= us * m01;
          }
        }
      }
    } else {
      const double i10 = inv[bpad], i11 = inv[2 * bpad];
      const double u0 = Uk[(size_t)(mslot * 2) * bpad], u1 = Uk[(size_t)(mslot * 2 + 1) * bpad];
      const double m00 = i00 * u0 + i10 * u1, m10 = i10 * u0 + i11 * u1;
      if (w == 1) {
        for (int ri = run0; ri < run1; ++ri) {
          const int* run = g.runs + 3 * (size_t)ri;
          const int rs = run[0], rd = run[1], len = run[2];
          for (int j = 0; j < len; ++j) {
            const double us0 = Uk[(size_t)((rs + j) * 2) * bpad], us1 = Uk[(size_t)((rs + j) * 2 + 1) * bpad];
            acc[(rd + j) * 64 + lane] -= us0 * m00 + us1 * m10;
          }
        }
      } else {
        const double v0 = Uk[(size_t)((mslot + 1) * 2) * bpad], v1 = Uk[(size_t)((mslot + 1) * 2 + 1) * bpad];
        const double m01 = i00 * v0 + i10 * v1, m11 = i10 * v0 + i11 * v1;
        for (int ri = run0; ri < run1; ++ri) {
          const int* run = g.runs + 3 * (size_t)ri;
          const int rs = run[0], rd = run[1], len = run[2];
          for (int j = 0; j < len; ++j) {
            const double us0 = Uk[(size_t)((rs + j) * 2) * bpad], us1 = Uk[(size_t)((rs + j) * 2 + 1) * bpad];
            acc[((rd + j) * 2) * 64 + lane] -= us0 * m00 + us1 * m10;
            acc[((rd + j) * 2 + 1) * 64 + lane] -= us0 * m01 + us1 * m11;
          }
        }
      }
    }
  }
  if (r0 == 0) {
    double colmax = fabs(d0);
    if (w == 2) colmax = fmax(colmax, fmax(fabs(d1), fabs(d2)));
    for (int i = w * w; i < nacc; ++i) colmax = fmax(colmax, fabs(acc[i * 64 + lane]));
    const double a = acc[lane];
    const double bb = (w == 2) ? acc[2 * 64 + lane] : 0.0;
    const double c = (w == 2) ? acc[3 * 64 + lane] : 0.0;
    const pp::PivotResult pr = pp::invert_pivot(w, a, bb, c, colmax, eps);
    double* invp = g.Dinv + (size_t)3 * p * bpad + b;
    invp[0] = pr.i00; invp[bpad] = pr.i10; invp[2 * bpad] = pr.i11;
    g.codes[(size_t)p * bpad + b] = (unsigned char)pr.code;
  }
  for (int i = 0; i < nacc; ++i) Up[(size_t)(r0 * w + i) * bpad] = acc[i * 64 + lane];
}

// counters[0..2] += (pos, neg, zero) over all pivots and active instances
__global__ __launch_bounds__(256) void k_count_codes(const unsigned char* __restrict__ codes, int npiv, int batch,
                                                     int bpad, int* counters) {
  __shared__ int red[3][256];
  int pos = 0, neg = 0, zero = 0;
  const size_t total = (size_t)npiv * bpad;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int b = (int)(i % bpad);
    if (b < batch) {
      const int c = codes[i];
      pos += c & 3; neg += (c >> 2) & 3; zero += (c >> 4) & 3;
    }
  }
  red[0][threadIdx.x] = pos; red[1][threadIdx.x] = neg; red[2][threadIdx.x] = zero;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
      for (int q = 0; q < 3; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x < 3 && red[threadIdx.x][0] != 0) atomicAdd(&counters[threadIdx.x], red[threadIdx.x][0]);
}

// ------------------------------------------------------------------------------------------
// Schur tile: acc[8][8] in registers over all panels holding rows of both tile ranges,
// then summed over the 64 instances of the wave through LDS.
__global__ __launch_bounds__(64) void k_schur_tiles(GroupDev g) {
  __shared__ double red[64][65];
  const int lane = threadIdx.x;
  const int b = blockIdx.y * 64 + lane;
  const size_t bpad = (size_t)g.bpad;
  const int tile = blockIdx.x;
  double acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.0;
  for (int r = g.stile_ptr[tile]; r < g.stile_ptr[tile + 1]; ++r) {
    const int* rec = g.stile_rec + 17 * (size_t)r;
    const int p = rec[0];
    const int w = g.piv_w[p];
    const double* Up = g.U + (size_t)g.piv_uoff[p] * bpad + b;
    const double* inv = g.Dinv + (size_t)3 * p * bpad + b;
    const double i00 = inv[0];
    double wa0[8], wa1[8], ub0[8], ub1[8];
    if (w == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int sa = rec[1 + i], sb = rec[9 + i];
        wa0[i] = (sa >= 0) ? Up[(size_t)sa * bpad] * i00 : 0.0;
        ub0[i] = (sb >= 0) ? Up[(size_t)sb * bpad] : 0.0;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] -= wa0[i] * ub0[j];
    } else {
      const double i10 = inv[bpad], i11 = inv[2 * bpad];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int sa = rec[1 + i], sb = rec[9 + i];
        const double a0 = (sa >= 0) ? Up[(size_t)(sa * 2) * bpad] : 0.0;
        const double a1 = (sa >= 0) ? Up[(size_t)(sa * 2 + 1) * bpad] : 0.0;
        wa0[i] = a0 * i00 + a1 * i10;
        wa1[i] = a0 * i10 + a1 * i11;
        ub0[i] = (sb >= 0) ? Up[(size_t)(sb * 2) * bpad] : 0.0;
        ub1[i] = (sb >= 0) ? Up[(size_t)(sb * 2 + 1) * bpad] : 0.0;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] -= wa0[i] * ub0[j] + wa1[i] * ub1[j];
    }
  }
  const double mask = (b < g.batch) ? 1.0 : 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) red[i * 8 + j][lane] = acc[i][j] * mask;
  __syncthreads();
  double s = 0.0;
  for (int l = 0; l < 64; ++l) s += red[lane][l];
  g.Spart[((size_t)blockIdx.y * gridDim.x + tile) * 64 + lane] = s;
}

// S[ci][cj] += sum over chunks of the tile partials (both triangles of the dense S)
__global__ __launch_bounds__(64) void k_schur_reduce(GroupDev g, int ntiles, double* __restrict__ S) {
  const int lane = threadIdx.x, tile = blockIdx.x;
  double s = 0.0;
  for (int c = 0; c < g.nchunk; ++c) s += g.Spart[((size_t)c * ntiles + tile) * 64 + lane];
  const int ci = g.stile_a[tile] * 8 + (lane >> 3), cj = g.stile_b[tile] * 8 + (lane & 7);
  if (ci < g.nc && cj < g.nc && ci >= cj) {
    S[(size_t)ci + (size_t)cj * g.nc] += s;
    if (ci != cj) S[(size_t)cj + (size_t)ci * g.nc] += s;
  }
}

__global__ void k_write_tail(const int* counters, double* tail) {
  if (threadIdx.x == 0) {
    tail[0] = (double)counters[2];  // numerically zero pivots
    tail[1] = (double)counters[0];
    tail[2] = (double)counters[1];
    tail[3] = 0.0;
  }
}

// Sfac = S + Q (Q lower triangle authoritative, dense column-major; may be null)
__global__ __launch_bounds__(256) void k_add_q(const double* __restrict__ S, const double* __restrict__ Q,
                                               double* __restrict__ Sfac, int nc) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)nc * nc) return;
  const int i = (int)(idx % nc), j = (int)(idx / nc);
  double q = 0.0;
  if (Q) q = (i >= j) ? Q[(size_t)i + (size_t)j * nc] : Q[(size_t)j + (size_t)i * nc];
  Sfac[idx] = S[idx] + q;
}

// ------------------------------------------------------------------------------------------
// thread-team context of dense_bk.hpp for one workgroup
struct TeamCtx {
  double* sv;  // [BK_THREADS/64] + spare
  int* si;
  __device__ int tid() const { return threadIdx.x; }
  __device__ int nthreads() const { return blockDim.x; }
  __device__ void sync() { __syncthreads(); }
  __device__ void argmax(double v, int i, double* vmax, int* imax) {
    for (int off = 32; off > 0; off >>= 1) {
      const double ov = __shfl_xor(v, off);
      const int oi = __shfl_xor(i, off);
      if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
    const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sv[wv] = v; si[wv] = i; }
    __syncthreads();
    double bv = sv[0]; int bi = si[0];
    for (int q = 1; q < nw; ++q)
      if (sv[q] > bv || (sv[q] == bv && si[q] < bi)) { bv = sv[q]; bi = si[q]; }
    *vmax = bv; *imax = bi;
  }
  __device__ double maxval(double v) {
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sv[wv] = v;
    __syncthreads();
    double r = sv[0];
    for (int q = 1; q < nw; ++q) r = fmax(r, sv[q]);
    return r;
  }
  __device__ double sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sv[wv] = v;
    __syncthreads();
    double r = 0.0;
    for (int q = 0; q < nw; ++q) r += sv[q];
    return r;
  }
};

__global__ __launch_bounds__(BK_THREADS) void k_bk_factor(int n, double* A, int* ipiv, double* work, int* info) {
  __shared__ double sv[16];
  __shared__ int si[16];
  TeamCtx ctx{sv, si};
  pp::BkInfo bi;
  __shared__ pp::BkInfo sbi;
  pp::bk_factor(ctx, n, A, n, ipiv, work, &sbi, BK_EPS);
  if (threadIdx.x == 0) { info[0] = sbi.npos; info[1] = sbi.nneg; info[2] = sbi.nzero; }
  (void)bi;
}

// xc = S^-1 (rc + rs)
__global__ __launch_bounds__(BK_THREADS) void k_coupling_solve(int n, const double* A, const int* ipiv,
                                                               const double* rc, const double* rs, double* xc) {
  __shared__ double sv[16];
  __shared__ int si[16];
  TeamCtx ctx{sv, si};
  for (int i = threadIdx.x; i < n; i += blockDim.x) xc[i] = (rc ? rc[i] : 0.0) + rs[i];
  __syncthreads();
  pp::bk_solve(ctx, n, A, n, ipiv, xc);
}

// ------------------------------------------------------------------------------------------
// forward substitution of one level: z_p = inv(P_p) (b_p - sum_k U[p,k] z_k)
__global__ __launch_bounds__(64) void k_fwd_level(GroupDev g, int piv0) {
  const int lane = threadIdx.x;
  const int b = blockIdx.y * 64 + lane;
  const size_t bpad = (size_t)g.bpad;
  const int p = g.lvl_piv[piv0 + blockIdx.x];
  const int w = g.piv_w[p], p0 = g.piv_start[p];
  double y0 = g.rhsT[(size_t)g.perm[p0] * bpad + b];
  double y1 = (w == 2) ? g.rhsT[(size_t)g.perm[p0 + 1] * bpad + b] : 0.0;
  for (int s = g.sfwd_ptr[p]; s < g.sfwd_ptr[p + 1]; ++s) {
    const int k = g.sfwd_k[s], ms = g.sfwd_mslot[s];
    const int wk = g.piv_w[k], k0 = g.piv_start[k];
    const double* Uk = g.U + (size_t)g.piv_uoff[k] * bpad + b;
    const double z0 = g.W[(size_t)k0 * bpad + b];
    if (wk == 1) {
      y0 -= Uk[(size_t)ms * bpad] * z0;
      if (w == 2) y1 -= Uk[(size_t)(ms + 1) * bpad] * z0;
    } else {
      const double z1 = g.W[(size_t)(k0 + 1) * bpad + b];
      y0 -= Uk[(size_t)(ms * 2) * bpad] * z0 + Uk[(size_t)(ms * 2 + 1) * bpad] * z1;
      if (w == 2) y1 -= Uk[(size_t)((ms + 1) * 2) * bpad] * z0 + Uk[(size_t)((ms + 1) * 2 + 1) * bpad] * z1;
    }
  }
  const double* inv = g.Dinv + (size_t)3 * p * bpad + b;
  if (w == 1) {
    g.W[(size_t)p0 * bpad + b] = inv[0] * y0;
  } else {
    const double i00 = inv[0], i10 = inv[bpad], i11 = inv[2 * bpad];
    g.W[(size_t)p0 * bpad + b] = i00 * y0 + i10 * y1;
    g.W[(size_t)(p0 + 1) * bpad + b] = i10 * y0 + i11 * y1;
  }
}

// coupling row c: rspart[chunk][c] = - sum over active instances and panels of U[c,k] z_k
__global__ __launch_bounds__(64) void k_fwd_coupling(GroupDev g) {
  const int lane = threadIdx.x;
  const int b = blockIdx.y * 64 + lane;
  const size_t bpad = (size_t)g.bpad;
  const int c = blockIdx.x;
  double s = 0.0;
  for (int t = g.crow_ptr[c]; t < g.crow_ptr[c + 1]; ++t) {
    const int k = g.crow_k[t], slot = g.crow_slot[t];
    const int wk = g.piv_w[k], k0 = g.piv_start[k];
    const double* Uk = g.U + (size_t)g.piv_uoff[k] * bpad + b;
    if (wk == 1) s -= Uk[(size_t)slot * bpad] * g.W[(size_t)k0 * bpad + b];
    else s -= Uk[(size_t)(slot * 2) * bpad] * g.W[(size_t)k0 * bpad + b] +
              Uk[(size_t)(slot * 2 + 1) * bpad] * g.W[(size_t)(k0 + 1) * bpad + b];
  }
  if (b >= g.batch) s = 0.0;
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) g.rspart[(size_t)blockIdx.y * g.nc + c] = s;
}

__global__ __launch_bounds__(256) void k_rs_reduce(GroupDev g, double* __restrict__ rs) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= g.nc) return;
  double s = 0.0;
  for (int q = 0; q < g.nchunk; ++q) s += g.rspart[(size_t)q * g.nc + c];
  rs[c] += s;
}

// back substitution of one level: x_p = z_p - inv(P_p) sum_i U[i,p]^T x_i  (x_i = xc for coupling rows)
__global__ __launch_bounds__(64) void k_bwd_level(GroupDev g, int piv0, const double* __restrict__ xc) {
  const int lane = threadIdx.x;
  const int b = blockIdx.y * 64 + lane;
  const size_t bpad = (size_t)g.bpad;
  const int p = g.lvl_piv[piv0 + blockIdx.x];
  const int w = g.piv_w[p], p0 = g.piv_start[p];
  const int nr = g.piv_rowptr[p + 1] - g.piv_rowptr[p];
  const int* ri = g.rowidx + g.piv_rowptr[p];
  const double* Up = g.U + ((size_t)g.piv_uoff[p] + (size_t)w * w) * bpad + b;
  double g0 = 0.0, g1 = 0.0;
  if (w == 1) {
    for (int j = 0; j < nr; ++j) {
      const int r = ri[j];
      const double x = (r < g.n) ? g.W[(size_t)r * bpad + b] : xc[r - g.n];
      g0 += Up[(size_t)j * bpad] * x;
    }
    const double inv = g.Dinv[(size_t)3 * p * bpad + b];
    g.W[(size_t)p0 * bpad + b] -= inv * g0;
  } else {
    for (int j = 0; j < nr; ++j) {
      const int r = ri[j];
      const double x = (r < g.n) ? g.W[(size_t)r * bpad + b] : xc[r - g.n];
      g0 += Up[(size_t)(j * 2) * bpad] * x;
      g1 += Up[(size_t)(j * 2 + 1) * bpad] * x;
    }
    const double* inv = g.Dinv + (size_t)3 * p * bpad + b;
    const double i00 = inv[0], i10 = inv[bpad], i11 = inv[2 * bpad];
    g.W[(size_t)p0 * bpad + b] -= i00 * g0 + i10 * g1;
    g.W[(size_t)(p0 + 1) * bpad + b] -= i10 * g0 + i11 * g1;
  }
}

// ------------------------------------------------------------------------------------------
struct Group {
  pp::Plan plan;
  GroupDev dev;
  int batch = 0, nraw = 0;
  std::vector<int> can_ptr, can_idx;
  std::vector<void*> allocs;
  int ntiles = 0;
  double *raw_own = nullptr, *rhs_own = nullptr;
};

}  // namespace

struct pp_solver {
  int device = 0;
  hipStream_t stream = nullptr;
  int nc = 0;
  bool symbolic_done = false, numeric_done = false, schur_done = false;
  std::vector<Group*> groups;
  double *S = nullptr, *S_own = nullptr, *Sfac = nullptr, *Qd = nullptr, *work = nullptr;
  double *rs = nullptr, *rs_own = nullptr, *rcd = nullptr, *xc = nullptr;
  int *ipiv = nullptr, *bkinfo = nullptr, *counters = nullptr;
  double mem_factor = 1.0;
  std::string err;
  // optional phase timing (HIP events on the handle's stream)
  bool profile = false;
  hipEvent_t ev[PP_NPHASE + 1][2];
  bool ev_made = false;
  bool ev_used[PP_NPHASE] = {};
  double phase_ms[PP_NPHASE] = {};
  int phase_launches[PP_NPHASE] = {};
  int phase_calls[PP_NPHASE] = {};
};

namespace {

int fail(pp_handle h, int status, const std::string& msg) {
  if (h) h->err = msg;
  return status;
}

// phase bracket: records events only in profile mode; durations are harvested lazily
struct PhaseScope {
  pp_handle h; int ph;
  PhaseScope(pp_handle h_, int ph_, int launches) : h(h_), ph(ph_) {
    if (!h->profile) return;
    if (!h->ev_made) {
      for (int i = 0; i < PP_NPHASE; ++i) { (void)hipEventCreate(&h->ev[i][0]); (void)hipEventCreate(&h->ev[i][1]); }
      h->ev_made = true;
    }
    if (h->ev_used[ph]) {   // harvest the previous bracket of this phase
      (void)hipEventSynchronize(h->ev[ph][1]);
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, h->ev[ph][0], h->ev[ph][1]) == hipSuccess) h->phase_ms[ph] += ms;
      h->ev_used[ph] = false;
    }
    h->phase_launches[ph] += launches;
    h->phase_calls[ph] += 1;
    (void)hipEventRecord(h->ev[ph][0], h->stream);
  }
  ~PhaseScope() {
    if (!h->profile) return;
    (void)hipEventRecord(h->ev[ph][1], h->stream);
    h->ev_used[ph] = true;
  }
};

#define PP_HIP(call)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess) {                                                                            \
      return fail(h, e_ == hipErrorOutOfMemory ? 1 : 3, std::string(#call) + ": " + hipGetErrorString(e_)); \
    }                                                                                                  \
  } while (0)

template <class T>
int dev_alloc(pp_handle h, Group* g, T** out, size_t count) {
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
  if (e != hipSuccess) return fail(h, e == hipErrorOutOfMemory ? 1 : 3, std::string("hipMalloc: ") + hipGetErrorString(e));
  if (g) g->allocs.push_back(p);
  *out = (T*)p;
  return 0;
}

template <class T>
int dev_upload(pp_handle h, Group* g, const T** out, const std::vector<T>& v) {
  T* p = nullptr;
  int rc = dev_alloc(h, g, &p, v.size());
  if (rc) return rc;
  if (!v.empty()) PP_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = p;
  return 0;
}

void free_group(Group* g) {
  for (void* p : g->allocs) (void)hipFree(p);
  delete g;
}

void free_globals(pp_handle h) {
  for (void* p : {(void*)h->S_own, (void*)h->Sfac, (void*)h->Qd, (void*)h->work, (void*)h->rs_own, (void*)h->rcd,
                  (void*)h->xc, (void*)h->ipiv, (void*)h->bkinfo, (void*)h->counters})
    if (p) (void)hipFree(p);
  h->S = h->S_own = h->Sfac = h->Qd = h->work = h->rs = h->rs_own = h->rcd = h->xc = nullptr;
  h->ipiv = h->bkinfo = h->counters = nullptr;
}

}  // namespace

// ==========================================================================================
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

void pp_destroy(pp_handle h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  for (Group* g : h->groups) free_group(g);
  free_globals(h);
  if (h->ev_made)
    for (int i = 0; i < PP_NPHASE; ++i) { (void)hipEventDestroy(h->ev[i][0]); (void)hipEventDestroy(h->ev[i][1]); }
  delete h;
}

const char* pp_last_error(pp_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int pp_begin_symbolic(pp_handle h, int n_coupling) {
  if (!h) return 3;
  if (n_coupling < 0) return fail(h, 3, "negative coupling dimension");
  PP_HIP(hipSetDevice(h->device));
  PP_HIP(hipStreamSynchronize(h->stream));
  for (Group* g : h->groups) free_group(g);
  h->groups.clear();
  free_globals(h);
  h->nc = n_coupling;
  h->symbolic_done = h->numeric_done = h->schur_done = false;
  return 0;
}

int pp_add_group(pp_handle h, int n, int batch, int nnzK, const int32_t* rowK, const int32_t* colK, int nnzB,
                 const int32_t* rowB, const int32_t* colB, int nraw, const int32_t* can_ptr, const int32_t* can_idx,
                 const double* rep_vals, int* group_out) {
  if (!h) return 3;
  if (h->symbolic_done) return fail(h, 3, "pp_add_group after pp_end_symbolic");
  if (batch <= 0 || n <= 0 || nraw < 0) return fail(h, 3, "bad group dimensions");
  Group* g = new Group();
  pp::PlanOptions opt;
  int rc = pp::build_plan(n, h->nc, nnzK, rowK, colK, nnzB, rowB, colB, rep_vals, opt, g->plan);
  if (rc != 0) { std::string e = g->plan.error; delete g; return fail(h, rc, "symbolic analysis failed: " + e); }
  const int ncan = nnzK + nnzB;
  g->batch = batch; g->nraw = nraw;
  g->can_ptr.assign(can_ptr, can_ptr + ncan + 1);
  g->can_idx.assign(can_idx, can_idx + can_ptr[ncan]);
  for (int v : g->can_idx)
    if (v < 0 || v >= nraw) { delete g; return fail(h, 3, "canonical map points outside the raw vector"); }
  if (g->plan.usize >= (int64_t)1 << 31) { delete g; return fail(h, 1, "panel storage exceeds 2^31 entries per instance"); }
  h->groups.push_back(g);
  if (group_out) *group_out = (int)h->groups.size() - 1;
  return 0;
}

int pp_end_symbolic(pp_handle h) {
  if (!h) return 3;
  PP_HIP(hipSetDevice(h->device));
  const int nc = h->nc;
  for (Group* g : h->groups) {
    const pp::Plan& P = g->plan;
    GroupDev& d = g->dev;
    std::memset(&d, 0, sizeof(d));
    d.n = P.n; d.nc = nc; d.batch = g->batch; d.bpad = (g->batch + WAVE - 1) / WAVE * WAVE;
    d.nchunk = d.bpad / WAVE; d.npiv = P.npiv; d.nraw = g->nraw; d.usize = P.usize;
    int rc;
    std::vector<int> uoff(P.piv_uoff.begin(), P.piv_uoff.end());
    std::vector<int> ftask, fsrc, runs, srec;
    ftask.reserve(P.ftasks.size() * 5);
    for (auto& t : P.ftasks) { ftask.insert(ftask.end(), {t.piv, t.r0, t.r1, t.src0, t.src1}); }
    for (auto& s : P.fsrcs) { fsrc.insert(fsrc.end(), {s.k, s.mslot, s.run0, s.run1}); }
    for (auto& r : P.runs) { runs.insert(runs.end(), {r.src, r.dst, r.len}); }
    for (auto& r : P.stile_rec) {
      srec.push_back(r.piv);
      for (int q = 0; q < 8; ++q) srec.push_back(r.slotA[q]);
      for (int q = 0; q < 8; ++q) srec.push_back(r.slotB[q]);
    }
    // gather map U position -> raw entries
    std::vector<int> can_of_pos((size_t)P.usize, -1);
    for (int e = 0; e < P.ncan; ++e) can_of_pos[(size_t)P.pos_of_can[e]] = e;
    std::vector<int> asm_ptr((size_t)P.usize + 1, 0), asm_idx;
    for (int64_t pos = 0; pos < P.usize; ++pos) {
      int e = can_of_pos[(size_t)pos];
      if (e >= 0) asm_idx.insert(asm_idx.end(), g->can_idx.begin() + g->can_ptr[e], g->can_idx.begin() + g->can_ptr[e + 1]);
      asm_ptr[(size_t)pos + 1] = (int)asm_idx.size();
    }
    if ((rc = dev_upload(h, g, &d.piv_w, P.piv_w))) return rc;
    if ((rc = dev_upload(h, g, &d.piv_start, P.piv_start))) return rc;
    if ((rc = dev_upload(h, g, &d.piv_uoff, uoff))) return rc;
    if ((rc = dev_upload(h, g, &d.piv_rowptr, P.piv_rowptr))) return rc;
    if ((rc = dev_upload(h, g, &d.rowidx, P.rowidx))) return rc;
    if ((rc = dev_upload(h, g, &d.perm, P.perm))) return rc;
    if ((rc = dev_upload(h, g, &d.iperm, P.iperm))) return rc;
    if ((rc = dev_upload(h, g, &d.ftask, ftask))) return rc;
    if ((rc = dev_upload(h, g, &d.fsrc, fsrc))) return rc;
    if ((rc = dev_upload(h, g, &d.runs, runs))) return rc;
    if ((rc = dev_upload(h, g, &d.lvl_piv, P.lvl_piv))) return rc;
    if ((rc = dev_upload(h, g, &d.sfwd_ptr, P.sfwd_ptr))) return rc;
    if ((rc = dev_upload(h, g, &d.sfwd_k, P.sfwd_k))) return rc;
    if ((rc = dev_upload(h, g, &d.sfwd_mslot, P.sfwd_mslot))) return rc;
    if ((rc = dev_upload(h, g, &d.crow_ptr, P.crow_ptr))) return rc;
    if ((rc = dev_upload(h, g, &d.crow_k, P.crow_k))) return rc;
    if ((rc = dev_upload(h, g, &d.crow_slot, P.crow_slot))) return rc;
    if ((rc = dev_upload(h, g, &d.stile_a, P.stile_a))) return rc;
    if ((rc = dev_upload(h, g, &d.stile_b, P.stile_b))) return rc;
    if ((rc = dev_upload(h, g, &d.stile_ptr, P.stile_ptr))) return rc;
    if ((rc = dev_upload(h, g, &d.stile_rec, srec))) return rc;
    if ((rc = dev_upload(h, g, &d.asm_ptr, asm_ptr))) return rc;
    if ((rc = dev_upload(h, g, &d.asm_idx, asm_idx))) return rc;
    g->ntiles = (int)P.stile_a.size();
    const size_t bp = (size_t)d.bpad;
    if ((rc = dev_alloc(h, g, &d.raw, (size_t)g->batch * g->nraw))) return rc;
    g->raw_own = d.raw;
    if ((rc = dev_alloc(h, g, &d.rawT, (size_t)g->nraw * bp))) return rc;
    if ((rc = dev_alloc(h, g, &d.U, (size_t)P.usize * bp))) return rc;
    if ((rc = dev_alloc(h, g, &d.Dinv, (size_t)3 * P.npiv * bp))) return rc;
    if ((rc = dev_alloc(h, g, &d.W, (size_t)(P.n + nc) * bp))) return rc;
    if ((rc = dev_alloc(h, g, &d.rhs, (size_t)g->batch * P.n))) return rc;
    g->rhs_own = d.rhs;
    if ((rc = dev_alloc(h, g, &d.rhsT, (size_t)P.n * bp))) return rc;
    if ((rc = dev_alloc(h, g, &d.xout, (size_t)g->batch * P.n))) return rc;
    if ((rc = dev_alloc(h, g, &d.Spart, (size_t)d.nchunk * std::max(g->ntiles, 1) * 64))) return rc;
    if ((rc = dev_alloc(h, g, &d.rspart, (size_t)d.nchunk * std::max(nc, 1)))) return rc;
    if ((rc = dev_alloc(h, g, &d.codes, (size_t)P.npiv * bp))) return rc;
  }
  int rc;
  const size_t nn = (size_t)nc * nc;
  if ((rc = dev_alloc<double>(h, nullptr, &h->S_own, nn + 4))) return rc;
  if ((rc = dev_alloc<double>(h, nullptr, &h->Sfac, nn))) return rc;
  if ((rc = dev_alloc<double>(h, nullptr, &h->Qd, nn))) return rc;
  if ((rc = dev_alloc<double>(h, nullptr, &h->work, 2 * (size_t)nc))) return rc;
  if ((rc = dev_alloc<double>(h, nullptr, &h->rs_own, nc))) return rc;
  if ((rc = dev_alloc<double>(h, nullptr, &h->rcd, nc))) return rc;
  if ((rc = dev_alloc<double>(h, nullptr, &h->xc, nc))) return rc;
  if ((rc = dev_alloc<int>(h, nullptr, &h->ipiv, nc))) return rc;
  if ((rc = dev_alloc<int>(h, nullptr, &h->bkinfo, 4))) return rc;
  if ((rc = dev_alloc<int>(h, nullptr, &h->counters, 4))) return rc;
  h->S = h->S_own;
  h->rs = h->rs_own;
  PP_HIP(hipMemset(h->S, 0, (nn + 4) * sizeof(double)));
  PP_HIP(hipMemset(h->rs, 0, std::max<size_t>(nc, 1) * sizeof(double)));
  PP_HIP(hipMemset(h->bkinfo, 0, 4 * sizeof(int)));
  h->symbolic_done = true;
  return 0;
}

static Group* get_group(pp_handle h, int group) {
  if (!h || group < 0 || group >= (int)h->groups.size()) return nullptr;
  return h->groups[group];
}

int pp_upload_values(pp_handle h, int group, const double* raw, int on_device) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_upload_values: bad group or symbolic phase not finished");
  PP_HIP(hipSetDevice(h->device));
  const size_t bytes = (size_t)g->batch * g->nraw * sizeof(double);
  if (bytes == 0 || raw == g->dev.raw) return 0;
  PP_HIP(hipMemcpyAsync(g->dev.raw, raw, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
  return 0;
}

double* pp_raw_buffer(pp_handle h, int group) {
  Group* g = get_group(h, group);
  return (g && h->symbolic_done) ? g->dev.raw : nullptr;
}

int pp_numeric_local(pp_handle h) {
  if (!h || !h->symbolic_done) return fail(h, 3, "pp_numeric_local before symbolic factorization");
  PP_HIP(hipSetDevice(h->device));
  hipStream_t st = h->stream;
  const int nc = h->nc;
  PP_HIP(hipMemsetAsync(h->S, 0, ((size_t)nc * nc + 4) * sizeof(double), st));
  PP_HIP(hipMemsetAsync(h->counters, 0, 4 * sizeof(int), st));
  for (Group* g : h->groups) {
    const pp::Plan& P = g->plan;
    GroupDev& d = g->dev;
    {
      PhaseScope ps(h, 0, 2);
      if (d.nraw > 0)
        hipLaunchKernelGGL(k_transpose_in, dim3((d.nraw + 63) / 64, d.nchunk), dim3(256), 0, st, d.raw, d.rawT, d.batch,
                           d.nraw, d.bpad);
      hipLaunchKernelGGL(k_assemble, dim3((unsigned)((P.usize + 3) / 4), d.nchunk), dim3(256), 0, st, d);
    }
    {
      PhaseScope ps(h, 1, P.n_levels);
      const size_t lds = (size_t)P.opt.acc_doubles * 64 * sizeof(double);
      for (int l = 0; l < P.n_levels; ++l) {
        const int t0 = P.flevel_ptr[l], nt = P.flevel_ptr[l + 1] - t0;
        if (nt > 0) hipLaunchKernelGGL(k_factor_level, dim3(nt, d.nchunk), dim3(64), lds, st, d, t0, PIVOT_EPS);
      }
    }
    {
      PhaseScope ps(h, 2, 3);
      hipLaunchKernelGGL(k_count_codes, dim3(std::min(1024, (int)(((size_t)P.npiv * d.bpad + 255) / 256))), dim3(256),
                         0, st, d.codes, P.npiv, d.batch, d.bpad, h->counters);
      if (g->ntiles > 0) {
        hipLaunchKernelGGL(k_schur_tiles, dim3(g->ntiles, d.nchunk), dim3(64), 0, st, d);
        hipLaunchKernelGGL(k_schur_reduce, dim3(g->ntiles), dim3(64), 0, st, d, g->ntiles, h->S);
      }
    }
  }
  hipLaunchKernelGGL(k_write_tail, dim3(1), dim3(64), 0, st, h->counters, h->S + (size_t)nc * nc);
  PP_HIP(hipGetLastError());
  h->numeric_done = true;
  h->schur_done = false;
  return 0;
}

double* pp_schur_buffer(pp_handle h) { return (h && h->symbolic_done) ? h->S : nullptr; }

int pp_bind_schur_buffer(pp_handle h, double* dev_ptr) {
  if (!h || !h->symbolic_done) return fail(h, 3, "pp_bind_schur_buffer before symbolic factorization");
  h->S = dev_ptr ? dev_ptr : h->S_own;
  return 0;
}

int pp_factor_schur(pp_handle h, const double* Q_host) {
  if (!h || !h->numeric_done) return fail(h, 3, "pp_factor_schur before pp_numeric_local");
  PP_HIP(hipSetDevice(h->device));
  hipStream_t st = h->stream;
  const int nc = h->nc;
  const size_t nn = (size_t)nc * nc;
  if (nc > 0) {
    if (Q_host) PP_HIP(hipMemcpyAsync(h->Qd, Q_host, nn * sizeof(double), hipMemcpyHostToDevice, st));
    PhaseScope ps(h, 3, 2);
    hipLaunchKernelGGL(k_add_q, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, h->S, Q_host ? h->Qd : nullptr,
                       h->Sfac, nc);
    hipLaunchKernelGGL(k_bk_factor, dim3(1), dim3(BK_THREADS), 0, st, nc, h->Sfac, h->ipiv, h->work, h->bkinfo);
  } else {
    PP_HIP(hipMemsetAsync(h->bkinfo, 0, 4 * sizeof(int), st));
  }
  PP_HIP(hipGetLastError());
  h->schur_done = true;
  return 0;
}

int pp_get_status(pp_handle h, int64_t out[4]) {
  if (!h || !h->schur_done) return fail(h, 3, "pp_get_status before pp_factor_schur");
  PP_HIP(hipSetDevice(h->device));
  double tail[4];
  int bk[4];
  PP_HIP(hipMemcpyAsync(tail, h->S + (size_t)h->nc * h->nc, sizeof(tail), hipMemcpyDeviceToHost, h->stream));
  PP_HIP(hipMemcpyAsync(bk, h->bkinfo, sizeof(bk), hipMemcpyDeviceToHost, h->stream));
  PP_HIP(hipStreamSynchronize(h->stream));
  const int64_t zero_blocks = (int64_t)(tail[0] + 0.5);
  out[1] = (int64_t)(tail[1] + 0.5) + bk[0];
  out[2] = (int64_t)(tail[2] + 0.5) + bk[1];
  out[3] = zero_blocks + bk[2];
  out[0] = (out[3] > 0) ? 2 : 0;
  return 0;
}

int pp_get_schur(pp_handle h, double* S_host) {
  if (!h || !h->numeric_done) return fail(h, 3, "pp_get_schur before pp_numeric_local");
  PP_HIP(hipSetDevice(h->device));
  PP_HIP(hipMemcpyAsync(S_host, h->S, (size_t)h->nc * h->nc * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  PP_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

int pp_upload_rhs(pp_handle h, int group, const double* rhs, int on_device) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_upload_rhs: bad group or symbolic phase not finished");
  PP_HIP(hipSetDevice(h->device));
  const size_t bytes = (size_t)g->batch * g->plan.n * sizeof(double);
  if (rhs == g->dev.rhs) return 0;
  PP_HIP(hipMemcpyAsync(g->dev.rhs, rhs, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
  return 0;
}

double* pp_rhs_buffer(pp_handle h, int group) {
  Group* g = get_group(h, group);
  return (g && h->symbolic_done) ? g->dev.rhs : nullptr;
}

int pp_solve_forward(pp_handle h) {
  if (!h || !h->numeric_done) return fail(h, 3, "pp_solve_forward before numeric factorization");
  PP_HIP(hipSetDevice(h->device));
  hipStream_t st = h->stream;
  const int nc = h->nc;
  PP_HIP(hipMemsetAsync(h->rs, 0, std::max<size_t>(nc, 1) * sizeof(double), st));
  for (Group* g : h->groups) {
    const pp::Plan& P = g->plan;
    GroupDev& d = g->dev;
    {
      PhaseScope ps(h, 4, P.n_levels + 1);
      hipLaunchKernelGGL(k_transpose_in, dim3((P.n + 63) / 64, d.nchunk), dim3(256), 0, st, d.rhs, d.rhsT, d.batch, P.n,
                         d.bpad);
      for (int l = 0; l < P.n_levels; ++l) {
        const int p0 = P.lvl_ptr[l], np = P.lvl_ptr[l + 1] - p0;
        if (np > 0) hipLaunchKernelGGL(k_fwd_level, dim3(np, d.nchunk), dim3(64), 0, st, d, p0);
      }
    }
    if (nc > 0) {
      PhaseScope ps(h, 5, 2);
      hipLaunchKernelGGL(k_fwd_coupling, dim3(nc, d.nchunk), dim3(64), 0, st, d);
      hipLaunchKernelGGL(k_rs_reduce, dim3((nc + 255) / 256), dim3(256), 0, st, d, h->rs);
    }
  }
  PP_HIP(hipGetLastError());
  return 0;
}

double* pp_rs_buffer(pp_handle h) { return (h && h->symbolic_done) ? h->rs : nullptr; }

int pp_bind_rs_buffer(pp_handle h, double* dev_ptr) {
  if (!h || !h->symbolic_done) return fail(h, 3, "pp_bind_rs_buffer before symbolic factorization");
  h->rs = dev_ptr ? dev_ptr : h->rs_own;
  return 0;
}

int pp_solve_coupling(pp_handle h, const double* rc_host) {
  if (!h || !h->schur_done) return fail(h, 3, "pp_solve_coupling before pp_factor_schur");
  PP_HIP(hipSetDevice(h->device));
  hipStream_t st = h->stream;
  const int nc = h->nc;
  if (nc == 0) return 0;
  if (rc_host) PP_HIP(hipMemcpyAsync(h->rcd, rc_host, (size_t)nc * sizeof(double), hipMemcpyHostToDevice, st));
  PhaseScope ps(h, 6, 1);
  hipLaunchKernelGGL(k_coupling_solve, dim3(1), dim3(BK_THREADS), 0, st, nc, h->Sfac, h->ipiv,
                     rc_host ? h->rcd : nullptr, h->rs, h->xc);
  PP_HIP(hipGetLastError());
  return 0;
}

int pp_solve_backward(pp_handle h) {
  if (!h || !h->schur_done) return fail(h, 3, "pp_solve_backward before pp_factor_schur");
  PP_HIP(hipSetDevice(h->device));
  hipStream_t st = h->stream;
  for (Group* g : h->groups) {
    const pp::Plan& P = g->plan;
    GroupDev& d = g->dev;
    PhaseScope ps(h, 7, P.n_levels + 1);
    for (int l = P.n_levels - 1; l >= 0; --l) {
      const int p0 = P.lvl_ptr[l], np = P.lvl_ptr[l + 1] - p0;
      if (np > 0) hipLaunchKernelGGL(k_bwd_level, dim3(np, d.nchunk), dim3(64), 0, st, d, p0, h->xc);
    }
    hipLaunchKernelGGL(k_transpose_out, dim3((P.n + 63) / 64, d.nchunk), dim3(256), 0, st, d.W, d.iperm, d.xout,
                       d.batch, P.n, d.bpad);
  }
  PP_HIP(hipGetLastError());
  return 0;
}

int pp_download_solution(pp_handle h, int group, double* x, int on_device) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_download_solution: bad group");
  PP_HIP(hipSetDevice(h->device));
  const size_t bytes = (size_t)g->batch * g->plan.n * sizeof(double);
  if (x != g->dev.xout)
    PP_HIP(hipMemcpyAsync(x, g->dev.xout, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
  if (!on_device) PP_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

double* pp_solution_buffer(pp_handle h, int group) {
  Group* g = get_group(h, group);
  return (g && h->symbolic_done) ? g->dev.xout : nullptr;
}

int pp_get_coupling_solution(pp_handle h, double* xc_host) {
  if (!h || !h->schur_done) return fail(h, 3, "pp_get_coupling_solution before pp_factor_schur");
  PP_HIP(hipSetDevice(h->device));
  if (h->nc > 0) PP_HIP(hipMemcpyAsync(xc_host, h->xc, (size_t)h->nc * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  PP_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

int pp_bind_raw_buffer(pp_handle h, int group, double* dev_ptr) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_bind_raw_buffer: bad group");
  g->dev.raw = dev_ptr ? dev_ptr : g->raw_own;
  return 0;
}

int pp_bind_rhs_buffer(pp_handle h, int group, double* dev_ptr) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_bind_rhs_buffer: bad group");
  g->dev.rhs = dev_ptr ? dev_ptr : g->rhs_own;
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
    ms_out[i] = h->phase_ms[i];
    launches_out[i] = h->phase_launches[i];
    calls_out[i] = h->phase_calls[i];
  }
  return 0;
}

int pp_increase_memory_allocation(pp_handle h, double factor) {
  if (!h) return 3;
  h->mem_factor *= factor;
  return 0;
}

int pp_synchronize(pp_handle h) {
  if (!h) return 3;
  PP_HIP(hipSetDevice(h->device));
  PP_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

int pp_group_stats(pp_handle h, int group, int64_t out[16]) {
  Group* g = get_group(h, group);
  if (!g) return fail(h, 3, "pp_group_stats: bad group");
  const pp::Plan& P = g->plan;
  const int64_t v[16] = {P.n, P.nc, g->batch, P.npiv, P.n_2x2, P.n_levels, P.nnz_L, P.usize, P.flops_factor,
                         P.flops_schur, (int64_t)P.ftasks.size(), (int64_t)P.runs.size(), (int64_t)P.stile_a.size(),
                         (int64_t)P.stile_rec.size(), P.ncan, g->nraw};
  std::memcpy(out, v, sizeof(v));
  return 0;
}

int pp_group_perm(pp_handle h, int group, int32_t* perm) {
  Group* g = get_group(h, group);
  if (!g) return fail(h, 3, "pp_group_perm: bad group");
  std::memcpy(perm, g->plan.perm.data(), sizeof(int) * g->plan.n);
  return 0;
}

}  // extern "C"
