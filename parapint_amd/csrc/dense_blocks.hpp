// Device building blocks shared by the dense S factorisation (dense.hip) and the diagonal blocks of the cyclic reduction
// (bcr.hip): thread-team contexts of dense_bk.hpp (Bunch-Kaufman) for a workgroup, the LDS-only barrier, and the 16 x 16
// diagonal-block / panel steps with DP-ALU DPP row broadcasts.
#pragma once
#include "common.hpp"

namespace {

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's outstanding GLOBAL stores
// (release fence at workgroup scope): in a loop that streams finished columns to global memory and never reads them
// back, that puts one global-store round trip (1-2 us) on the critical path of every barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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

// TeamCtx for work that lives entirely in LDS: the barriers order LDS traffic only (see lds_barrier: __syncthreads() would
// also wait for every outstanding global store, a memory round trip on the critical path of each of the ~10 barriers of a
// pivot step).
struct TeamCtxLds {
  double* sv;
  int* si;
  __device__ int tid() const { return threadIdx.x; }
  __device__ int nthreads() const { return blockDim.x; }
  __device__ void sync() { lds_barrier(); }
  __device__ void argmax(double v, int i, double* vmax, int* imax) {
    for (int off = 32; off > 0; off >>= 1) {
      const double ov = __shfl_xor(v, off);
      const int oi = __shfl_xor(i, off);
      if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
    const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    lds_barrier();
    if ((threadIdx.x & 63) == 0) { sv[wv] = v; si[wv] = i; }
    lds_barrier();
    double bv = sv[0]; int bi = si[0];
    for (int q = 1; q < nw; ++q)
      if (sv[q] > bv || (sv[q] == bv && si[q] < bi)) { bv = sv[q]; bi = si[q]; }
    *vmax = bv; *imax = bi;
  }
  __device__ double maxval(double v) {
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    lds_barrier();
    if ((threadIdx.x & 63) == 0) sv[wv] = v;
    lds_barrier();
    double r = sv[0];
    for (int q = 1; q < nw; ++q) r = fmax(r, sv[q]);
    return r;
  }
  __device__ double sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    lds_barrier();
    if ((threadIdx.x & 63) == 0) sv[wv] = v;
    lds_barrier();
    double r = 0.0;
    for (int q = 0; q < nw; ++q) r += sv[q];
    return r;
  }
};

// The same team interface for ONE wave: no workgroup barriers (LDS traffic of a wave is ordered; the fence keeps the
// compiler from moving accesses across the point), reductions by lane shuffles.
struct WaveCtx {
  __device__ int tid() const { return threadIdx.x & 63; }
  __device__ int nthreads() const { return 64; }
  __device__ void sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
  __device__ void argmax(double v, int i, double* vmax, int* imax) {
    for (int off = 32; off > 0; off >>= 1) {
      const double ov = __shfl_xor(v, off);
      const int oi = __shfl_xor(i, off);
      if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
    *vmax = v; *imax = i;
  }
  __device__ double maxval(double v) {
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
  }
  __device__ double sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
  }
};


// Broadcast inside a row of 16 lanes without leaving the vector unit: DP-ALU DPP (gfx90a+: 64-bit VOP1/VOP2 operations take
// row_newbcast:K = "operand 0 comes from lane K of my row of 16").  One v_fmac_f64_dpp replaces two v_readlane_b32 + a
// v_fma_f64 in the column updates of an in-register triangular factor / solve whose rows live one per lane (all four
// rows of 16 lanes of the wave holding the same 16 matrix rows).  The s_nop covers the two wait states a DPP read needs
// behind a VALU write of the same register (the hazard recogniser does not look into inline assembly).
template <int K>
__device__ __forceinline__ void fmac_row_bcast(double& acc, double from_lane_k, double mul) {      // acc += from_lane_k[K] * mul
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
               : "+v"(acc) : "v"(from_lane_k), "v"(mul), "n"(K));
}
template <int K>
__device__ __forceinline__ double mov_row_bcast(double from_lane_k) {
  double r;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(from_lane_k), "n"(K));
  return r;
}
// steps K.. of the unpivoted LDL^T of a 16 x 16 block, lane (mod 16) = row, row[] in registers (see k_ldl_regs (b))
template <int K>
struct DiagSteps {
  static __device__ __forceinline__ void run(double (&row)[16], double& d, double& rd, int& bad, int& signs, int nb,
                                             double eps_anorm, double anorm, double* dl, double* rdl, int lane) {
    const double lik = row[K] * rd;
    const double nlik = -lik;
    double dn = 1.0, rdn = 1.0;
    if constexpr (K + 1 < 16) {
      fmac_row_bcast<K>(row[K + 1], row[K + 1], nlik);
      dn = mov_row_bcast<K + 1>(row[K + 1]);
      if (K + 1 < nb) {
        if (!(fabs(dn) > eps_anorm)) { bad = 1; dn = (anorm > 0.0 ? anorm : 1.0); }
        signs |= (dn > 0.0) ? 1 : 2;
      } else {
        dn = 1.0;
      }
      rdn = fast_rcp(dn);
    }
#pragma unroll
    for (int j = K + 2; j < 16; ++j) fmac_row_bcast<K>(row[j], row[j], nlik);
    row[K] = lik;
    if (lane == 0) { dl[K] = d; rdl[K] = rd; }
    d = dn; rd = rdn;
    if constexpr (K + 1 < 16) DiagSteps<K + 1>::run(row, d, rd, bad, signs, nb, eps_anorm, anorm, dl, rdl, lane);
  }
};
// W = A21 L11^{-T}, thread = row of A21 (wrow), L11 rows one per lane of every row of 16 lanes (lrow): column J
template <int J, int K>
struct PanelSolve {
  static __device__ __forceinline__ void run(double (&wrow)[16], const double (&lrow)[16], double nwj) {
    if constexpr (K < 16) {
      fmac_row_bcast<K>(wrow[K], lrow[J], nwj);        // wrow[K] -= wrow[J] * L11[K][J]
      PanelSolve<J, K + 1>::run(wrow, lrow, nwj);
    }
  }
};
template <int J>
struct PanelSolveCols {
  static __device__ __forceinline__ void run(double (&wrow)[16], const double (&lrow)[16]) {
    if constexpr (J + 1 < 16) {
      PanelSolve<J, J + 1>::run(wrow, lrow, -wrow[J]);
      PanelSolveCols<J + 1>::run(wrow, lrow);
    }
  }
};


}  // namespace
