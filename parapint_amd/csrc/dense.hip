// Dense S (n_c x n_c, replicated on every rank): optimistic LDL^T on the fp64 matrix cores, Bunch-Kaufman fallback, coupling
// solve (mpi_explicit_schur_complement.py:347-361, 388-391).
#include "common.hpp"
#include "dense_blocks.hpp"

namespace {

// Sfac = S + Q (Q lower triangle authoritative, dense column-major; may be null)
__global__ __launch_bounds__(256) void k_add_q(const double* __restrict__ S, const double* __restrict__ Q,
                                               double* __restrict__ Sfac, double* __restrict__ Sldl, int nc) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)nc * nc) return;
  const int i = (int)(idx % nc), j = (int)(idx / nc);
  double q = 0.0;
  if (Q) q = (i >= j) ? Q[(size_t)i + (size_t)j * nc] : Q[(size_t)j + (size_t)i * nc];
  const double v = S[idx] + q;
  Sfac[idx] = v;
  Sldl[idx] = v;
}

// ------------------------------------------------------------------------------------------
// Optimistic dense LDL^T of S without pivoting, blocked (panel width 32), one workgroup.
// The trailing updates run on the fp64 matrix cores (v_mfma_f64_16x16x4_f64): S is a genuine
// dense symmetric panel.  The result is accepted only if every pivot has the same sign (S
// definite, where the unpivoted factorisation is unconditionally stable) and no pivot is
// numerically zero; otherwise mode[0] stays 0 and k_bk_factor (Bunch-Kaufman) takes over on the
// untouched copy.  At the points the interior-point method accepts, S of a stochastic program
// is positive definite (Haynsworth: every K_i carries its own negative eigenvalues).
constexpr int LDL_NB = 32;
constexpr int LDL_THREADS = 512;

__global__ __launch_bounds__(LDL_THREADS) void k_ldl_blocked(int n, double* __restrict__ A, double* __restrict__ dvec,
                                                             int* __restrict__ mode, int* __restrict__ info, double eps) {
  __shared__ double Db[LDL_NB][LDL_NB + 1];
  __shared__ double dl[LDL_NB];      // pivots of the current panel
  __shared__ double red[LDL_THREADS / 64];
  __shared__ int sflags[2];          // [0] bad pivot seen, [1] sign bookkeeping (bit0 pos, bit1 neg)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nwv = LDL_THREADS / 64;
  const size_t lda = (size_t)n;
  // scale for the zero-pivot test: max |diagonal|
  double loc = 0.0;
  for (int i = tid; i < n; i += LDL_THREADS) loc = fmax(loc, fabs(A[i + i * lda]));
  for (int off = 32; off > 0; off >>= 1) loc = fmax(loc, __shfl_xor(loc, off));
  if (lane == 0) red[wv] = loc;
  if (tid == 0) { sflags[0] = 0; sflags[1] = 0; }
  __syncthreads();
  double anorm = 0.0;
  for (int q = 0; q < nwv; ++q) anorm = fmax(anorm, red[q]);
  for (int j0 = 0; j0 < n; j0 += LDL_NB) {
    const int nb = min(LDL_NB, n - j0), j1 = j0 + nb, m = n - j1;
    // (1) diagonal block -> LDS, factored by wave 0 (lane = row; LDS ops of one wave are in order)
    for (int idx = tid; idx < nb * nb; idx += LDL_THREADS) {
      const int i = idx % nb, j = idx / nb;
      Db[i][j] = (i >= j) ? A[(j0 + i) + (size_t)(j0 + j) * lda] : 0.0;
    }
    __syncthreads();
    if (wv == 0) {
      // lane = row of the 32x32 block, the row lives in registers; column k is broadcast with shuffles
      double row[LDL_NB];
      const int i = lane & 31;
#pragma unroll
      for (int j = 0; j < LDL_NB; ++j) row[j] = (i < nb && j < nb) ? Db[i][j] : ((i == j) ? 1.0 : 0.0);
      int bad = 0, signs = 0;
#pragma unroll
      for (int k = 0; k < LDL_NB; ++k) {
        const double colk = row[k];
        double d = bcastd(colk, k);
        if (k < nb) {
          if (!(fabs(d) > eps * anorm)) { bad = 1; d = (anorm > 0.0 ? anorm : 1.0); }
          signs |= (d > 0.0) ? 1 : 2;
        }
        const double lik = colk * fast_rcp(d);
#pragma unroll
        for (int j = k + 1; j < LDL_NB; ++j) {
          const double ajk = bcastd(colk, j);
          if (i >= j) row[j] -= lik * ajk;
        }
        if (i > k) row[k] = lik;
        else if (i == k) row[k] = d;
        __builtin_amdgcn_sched_barrier(0);   // keep the broadcasts of later columns from being hoisted (SGPR pressure)
      }
      if (lane < nb) {
#pragma unroll
        for (int j = 0; j < LDL_NB; ++j) if (j < nb) Db[lane][j] = row[j];
      }
      if (lane == 0) { if (bad) sflags[0] = 1; sflags[1] |= signs; }
    }
    __syncthreads();
    if (tid < nb) dl[tid] = Db[tid][tid];
    __syncthreads();
    // write the factored diagonal block back (unit lower L11, pivots on the diagonal)
    for (int idx = tid; idx < nb * nb; idx += LDL_THREADS) {
      const int i = idx % nb, j = idx / nb;
      if (i > j) A[(j0 + i) + (size_t)(j0 + j) * lda] = Db[i][j];
      else if (i == j) { A[(j0 + i) + (size_t)(j0 + j) * lda] = dl[i]; dvec[j0 + i] = dl[i]; }
    }
    // (2) panel: W = A21 L11^{-T} row by row (thread = row), then L21 = W D^{-1}, stored in A
    if (nb == LDL_NB) {
      for (int r = tid; r < m; r += LDL_THREADS) {
        double wrow[LDL_NB];
#pragma unroll
        for (int k = 0; k < LDL_NB; ++k) wrow[k] = A[(j1 + r) + (size_t)(j0 + k) * lda];   // all loads in flight at once
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < LDL_NB; ++k) {
          double v = wrow[k];
#pragma unroll
          for (int j = 0; j < k; ++j) v -= wrow[j] * Db[k][j];
          wrow[k] = v;
          __builtin_amdgcn_sched_barrier(0);   // keep the LDS reads of later columns from being hoisted (register pressure)
        }
#pragma unroll
        for (int k = 0; k < LDL_NB; ++k) A[(j1 + r) + (size_t)(j0 + k) * lda] = wrow[k] * fast_rcp(dl[k]);
      }
    } else {
      for (int r = tid; r < m; r += LDL_THREADS) {   // ragged last panel: W kept in place, scaled afterwards
        for (int k = 0; k < nb; ++k) {
          double v = A[(j1 + r) + (size_t)(j0 + k) * lda];
          for (int j = 0; j < k; ++j) v -= A[(j1 + r) + (size_t)(j0 + j) * lda] * Db[k][j];
          A[(j1 + r) + (size_t)(j0 + k) * lda] = v;
        }
        for (int k = 0; k < nb; ++k) A[(j1 + r) + (size_t)(j0 + k) * lda] /= dl[k];
      }
    }
    __syncthreads();
    // (3) trailing update A22 -= L21 D L21^T on 16x16 tiles with fp64 MFMA (lower tiles only)
    if (m > 0) {
      const int nt = (m + 15) / 16;
      const int ntiles = nt * (nt + 1) / 2;
      const int li = lane & 15, lk = lane >> 4;
      for (int tix = wv; tix < ntiles; tix += nwv) {
        // tile index -> (I >= J)
        int I = (int)((sqrt(8.0 * tix + 1.0) - 1.0) * 0.5);
        while ((I + 1) * (I + 2) / 2 <= tix) ++I;
        while (I * (I + 1) / 2 > tix) --I;
        const int J = tix - I * (I + 1) / 2;
        const int ra = j1 + 16 * I + li, rb = j1 + 16 * J + li;
        const bool va = ra < n, vb = rb < n;
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
        double av[LDL_NB / 4], bv[LDL_NB / 4];
#pragma unroll
        for (int q = 0; q < LDL_NB / 4; ++q) {     // all operand loads first: independent, coalesced
          const int k = 4 * q + lk;
          const bool vk = k < nb;
          av[q] = (va && vk) ? A[ra + (size_t)(j0 + k) * lda] * dl[k] : 0.0;
          bv[q] = (vb && vk) ? A[rb + (size_t)(j0 + k) * lda] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < LDL_NB / 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc, 0, 0, 0);
        const int col = j1 + 16 * J + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = j1 + 16 * I + lk + 4 * r;
          if (row < n && col < n && row >= col) A[row + (size_t)col * lda] -= acc[r];
        }
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    const bool ok = (sflags[0] == 0) && (sflags[1] == 1 || sflags[1] == 2 || n == 0);
    mode[0] = ok ? 1 : 0;
    if (ok) { info[0] = (sflags[1] == 1) ? n : 0; info[1] = (sflags[1] == 2) ? n : 0; info[2] = 0; }
  }
}

// Multi-workgroup variant for large n (the 1000 x 1000 S of configuration C5): the same blocked algorithm with
// the panel and the trailing update spread over the chip, two launches per 32-column panel.
//   k_dense_anorm   scale of the zero-pivot test (max |diagonal|), clears the acceptance flags
//   k_dense_panel   every workgroup factors the 32 x 32 diagonal block redundantly in LDS / registers (it is
//                   tiny) and solves 256 rows of the panel against it; workgroup 0 stores the block and the flags
//   k_dense_update  A22 -= L21 D L21^T, one 16 x 16 tile per wave on the fp64 matrix cores
//   k_dense_finish  acceptance rule of k_ldl_blocked -> mode / inertia counters
constexpr int DN_THREADS = 256;

__global__ __launch_bounds__(256) void k_dense_anorm(int n, const double* __restrict__ A, double* __restrict__ anorm,
                                                     int* __restrict__ flags) {
  __shared__ double red[4];
  double loc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) loc = fmax(loc, fabs(A[i + (size_t)i * n]));
  for (int off = 32; off > 0; off >>= 1) loc = fmax(loc, __shfl_xor(loc, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = loc;
  __syncthreads();
  if (threadIdx.x == 0) {
    anorm[0] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    flags[0] = 0; flags[1] = 0;
  }
}

__global__ __launch_bounds__(DN_THREADS) void k_dense_panel(int n, double* __restrict__ A, double* __restrict__ dvec,
                                                            const double* __restrict__ anorm_p, int* __restrict__ flags,
                                                            double* __restrict__ stage, int j0, double eps) {
  __shared__ double Db[LDL_NB][LDL_NB + 1];
  __shared__ double dl[LDL_NB];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const size_t lda = (size_t)n;
  const int nb = min(LDL_NB, n - j0), j1 = j0 + nb, m = n - j1;
  const double anorm = anorm_p[0];
  for (int idx = tid; idx < nb * nb; idx += DN_THREADS) {
    const int i = idx % nb, j = idx / nb;
    Db[i][j] = (i >= j) ? A[(j0 + i) + (size_t)(j0 + j) * lda] : 0.0;
  }
  // this workgroup's row of the panel: requested before the diagonal block is factored
  const int r = (int)(blockIdx.x - 1) * DN_THREADS + tid;
  const bool have_row = blockIdx.x > 0 && r < m && nb == LDL_NB;
  double wrow[LDL_NB];
#pragma unroll
  for (int k = 0; k < LDL_NB; ++k) wrow[k] = have_row ? A[(j1 + r) + (size_t)(j0 + k) * lda] : 0.0;
  __syncthreads();
  if (wv == 0) {
    double row[LDL_NB];
    const int i = lane & 31;
#pragma unroll
    for (int j = 0; j < LDL_NB; ++j) row[j] = (i < nb && j < nb) ? Db[i][j] : ((i == j) ? 1.0 : 0.0);
    int bad = 0, signs = 0;
#pragma unroll
    for (int k = 0; k < LDL_NB; ++k) {
      const double colk = row[k];
      double d = bcastd(colk, k);
      if (k < nb) {
        if (!(fabs(d) > eps * anorm)) { bad = 1; d = (anorm > 0.0 ? anorm : 1.0); }
        signs |= (d > 0.0) ? 1 : 2;
      }
      const double lik = colk * fast_rcp(d);    // (no IEEE division sequence in the 32-step pivot chain)
#pragma unroll
      for (int j = k + 1; j < LDL_NB; ++j) {
        const double ajk = bcastd(colk, j);
        if (i >= j) row[j] -= lik * ajk;
      }
      if (i > k) row[k] = lik;
      else if (i == k) row[k] = d;
      __builtin_amdgcn_sched_barrier(0);
    }
    if (lane < nb) {
#pragma unroll
      for (int j = 0; j < LDL_NB; ++j) if (j < nb) Db[lane][j] = row[j];
    }
    if (lane == 0 && blockIdx.x == 0) {
      if (bad) atomicOr(&flags[0], 1);
      atomicOr(&flags[1], signs);
    }
  }
  __syncthreads();
  if (tid < nb) dl[tid] = Db[tid][tid];
  __syncthreads();
  if (blockIdx.x == 0) {
    // factored diagonal block (unit lower L11, pivots on the diagonal): the other workgroups of this launch still
    // read the unfactored block from A, so it goes to a staging tile and k_dense_update copies it in; only the last
    // panel (no other workgroup, no update launch) is stored directly
    for (int idx = tid; idx < nb * nb; idx += DN_THREADS) {
      const int i = idx % nb, j = idx / nb;
      const double v = (i > j) ? Db[i][j] : ((i == j) ? dl[i] : 0.0);
      if (m > 0) stage[idx] = v;
      else if (i >= j) A[(j0 + i) + (size_t)(j0 + j) * lda] = v;
    }
    if (tid < nb) dvec[j0 + tid] = dl[tid];
    return;
  }
  if (have_row) {          // W = A21 L11^{-T}, L21 = W D^{-1}
#pragma unroll
    for (int k = 0; k < LDL_NB; ++k) {
      double v = wrow[k];
#pragma unroll
      for (int j = 0; j < k; ++j) v -= wrow[j] * Db[k][j];
      wrow[k] = v;
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < LDL_NB; ++k) A[(j1 + r) + (size_t)(j0 + k) * lda] = wrow[k] * fast_rcp(dl[k]);
  }
}

__global__ __launch_bounds__(DN_THREADS) void k_dense_update(int n, double* __restrict__ A, const double* __restrict__ dvec,
                                                             const double* __restrict__ stage, int j0) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t lda = (size_t)n;
  if (blockIdx.x == 0) {   // the factored diagonal block of this panel (staged by k_dense_panel) -> A
    for (int idx = threadIdx.x; idx < LDL_NB * LDL_NB; idx += DN_THREADS) {
      const int i = idx % LDL_NB, j = idx / LDL_NB;
      if (i >= j) A[(j0 + i) + (size_t)(j0 + j) * lda] = stage[idx];
    }
  }
  const int j1 = j0 + LDL_NB, m = n - j1;
  const int nt = (m + 15) / 16, ntiles = nt * (nt + 1) / 2;
  const int tix = (int)blockIdx.x * (DN_THREADS / 64) + wv;
  if (tix >= ntiles) return;
  const int li = lane & 15, lk = lane >> 4;
  int I = (int)((sqrt(8.0 * tix + 1.0) - 1.0) * 0.5);
  while ((I + 1) * (I + 2) / 2 <= tix) ++I;
  while (I * (I + 1) / 2 > tix) --I;
  const int J = tix - I * (I + 1) / 2;
  const int ra = j1 + 16 * I + li, rb = j1 + 16 * J + li;
  const bool va = ra < n, vb = rb < n;
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  double av[LDL_NB / 4], bv[LDL_NB / 4];
#pragma unroll
  for (int q = 0; q < LDL_NB / 4; ++q) {
    const int k = 4 * q + lk;
    av[q] = va ? A[ra + (size_t)(j0 + k) * lda] * dvec[j0 + k] : 0.0;
    bv[q] = vb ? A[rb + (size_t)(j0 + k) * lda] : 0.0;
  }
#pragma unroll
  for (int q = 0; q < LDL_NB / 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc, 0, 0, 0);
  const int col = j1 + 16 * J + li;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = j1 + 16 * I + lk + 4 * r;
    if (row < n && col < n && row >= col) A[row + (size_t)col * lda] -= acc[r];
  }
}

__global__ void k_dense_finish(int n, const int* __restrict__ flags, int* __restrict__ mode, int* __restrict__ info) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const bool ok = (flags[0] == 0) && (flags[1] == 1 || flags[1] == 2 || n == 0);
    mode[0] = ok ? 1 : 0;
    if (ok) { info[0] = (flags[1] == 1) ? n : 0; info[1] = (flags[1] == 2) ? n : 0; info[2] = 0; }
  }
}


// Register-resident variant for n <= 16 * LDLR_NT (= 208; the reference configurations have n_c = 200):
// the whole lower triangle lives in the MFMA accumulators of the 8 waves (91 tiles of 16x16, <= 12 per
// wave) for the entire factorisation, so a trailing update is LDS reads + fp64 MFMAs only -- no global
// read-modify-write round trips inside the panel loop.  Per 16-column panel (= one tile column): its
// tiles go to LDS, wave 0 factors the 16x16 diagonal block in registers (column broadcasts inside the rows of 16
// lanes by DP-ALU DPP, see fmac_row_bcast), one thread per row solves the panel against it, the finished columns are streamed to global
// memory (stores only), and every wave updates the tiles it still owns.  Same acceptance rule and
// output format as k_ldl_blocked.
constexpr int LDLR_NT = 13;
constexpr int LDLR_TPW = 12;  // 8 waves * 12 >= 91 tiles
constexpr int LDLR_NB = 16;
constexpr int LDLR_LD = 18;   // LDS row stride in doubles: conflict-free MFMA operand reads

// BLOCK: the same kernel on a diagonal block (n <= 208 columns) of a larger matrix, in place: A points at the block's
// first element, lda_in is the leading dimension of the whole matrix, the lower triangle is read (the trailing updates of
// the blocked algorithm below keep only that one current), the scale of the zero-pivot test comes from the whole matrix
// (anorm_p) and the acceptance flags are merged into gflags (k_dense_finish decides).
template <bool DPP, bool BLOCK = false>
__global__ __launch_bounds__(LDL_THREADS) void k_ldl_regs(int n, const double* __restrict__ S, const double* __restrict__ Q,
                                                          double* __restrict__ A, double* __restrict__ dvec,
                                                          int* __restrict__ mode, int* __restrict__ info, double eps,
                                                          int lda_in = 0, const double* __restrict__ anorm_p = nullptr,
                                                          int* __restrict__ gflags = nullptr) {
  __shared__ double P[16 * LDLR_NT][LDLR_LD];
  __shared__ double dl[LDLR_NB], rdl[LDLR_NB];
  __shared__ double red[LDL_THREADS / 64];
  __shared__ int sflags[2];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = LDL_THREADS / 64;
  const int li = lane & 15, lk = lane >> 4;
  const size_t lda = BLOCK ? (size_t)lda_in : (size_t)n;
  const int nt = (n + 15) / 16, ntt = nt * (nt + 1) / 2;
  // the input is S + Q (Q: lower triangle authoritative, may be null), read straight from the all-reduced buffer:
  // no separate add/copy kernel in front of the factorisation; S itself stays untouched for the pivoted fallback
  double loc = 0.0;
  if (!BLOCK)
    for (int i = tid; i < n; i += LDL_THREADS) loc = fmax(loc, fabs(S[i + i * lda] + (Q ? Q[i + i * lda] : 0.0)));
  // tiles of this wave: t = wv + 8 s  <->  (I >= J), t = I (I + 1) / 2 + J
  double4_t acc[LDLR_TPW];
  int tIJ[LDLR_TPW];   // wave-uniform (SGPR): I << 8 | J, or -1
#pragma unroll
  for (int s = 0; s < LDLR_TPW; ++s) {
    const int t = wv + nwv * s;
    int I = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((I + 1) * (I + 2) / 2 <= t) ++I;
    while (I * (I + 1) / 2 > t) --I;
    const int J = t - I * (I + 1) / 2;
    tIJ[s] = (t < ntt) ? ((I << 8) | J) : -1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * I + lk + 4 * r, col = 16 * J + li;
      double v = 0.0;
      if (t < ntt && row < n && col < n) {
        // S is symmetric in memory (both triangles are written): element (col, row) instead of (row, col) makes the 16
        // lanes of a row of the wave read 128 contiguous bytes instead of 16 cache lines
        if (BLOCK) {
          v = (row >= col) ? A[row + (size_t)col * lda] : A[col + (size_t)row * lda];
        } else {
          v = S[col + (size_t)row * lda];
          if (Q) v += (row >= col) ? Q[row + (size_t)col * lda] : Q[col + (size_t)row * lda];
        }
      }
      acc[s][r] = v;
    }
  }
  for (int off = 32; off > 0; off >>= 1) loc = fmax(loc, __shfl_xor(loc, off));
  if (lane == 0) red[wv] = loc;
  if (tid == 0) { sflags[0] = 0; sflags[1] = 0; }
  __syncthreads();
  double anorm = 0.0;
  for (int q = 0; q < nwv; ++q) anorm = fmax(anorm, red[q]);
  if (BLOCK) anorm = anorm_p[0];
  for (int jt_loop = 0; jt_loop < nt; ++jt_loop) {
    // panel index and per-lane tile coordinates behind optimisation barriers: otherwise the LDS addresses of
    // all 12 tiles become loop-carried induction variables / hoisted invariants and pin ~100 registers
    int jt = jt_loop, liv = li, lkv = lk;
    asm volatile("" : "+s"(jt), "+v"(liv), "+v"(lkv));
    const int j0 = 16 * jt;
    const int nb = min(LDLR_NB, n - j0), j1 = j0 + nb, m = n - j1;
    // (a) the panel's tiles (tile column jt): accumulators -> LDS (rows relative to j0)
#pragma unroll
    for (int s = 0; s < LDLR_TPW; ++s) {
      if (tIJ[s] >= 0 && (tIJ[s] & 255) == jt) {
        const int r0 = 16 * ((tIJ[s] >> 8) - jt);
#pragma unroll
        for (int r = 0; r < 4; ++r) P[r0 + lkv + 4 * r][liv] = acc[s][r];
      }
    }
    lds_barrier();
    // (b) diagonal block by wave 0: lane = row (16 rows, lanes 16.. duplicate them), the row lives in
    // registers.  Column k is broadcast with v_readlane from lane k (symmetry: see below) -- an LDS broadcast
    // costs two ~130-cycle round trips per step.  Pivot k + 1 is final as soon as step k has updated column
    // k + 1: its reciprocal (v_rcp_f64 + two Newton steps) is started first and overlaps the rest of the
    // step.  Mask-free: in a ragged last panel rows/columns >= nb carry don't-care values that never reach a
    // stored result; the unit diagonal is implicit.
    int wv_here = wv;
    asm volatile("" : "+s"(wv_here));   // opaque: keeps the panel loop from being unswitched on the wave id (two copies
                                        // of the loop double the accumulator live ranges and spill them)
    if (wv_here == 0) {
      double row[LDLR_NB];
      const int i = lane & 15;
#pragma unroll
      for (int j = 0; j < LDLR_NB; ++j) row[j] = P[i][j];
      int bad = 0, signs = 0;
      double d = DPP ? mov_row_bcast<0>(row[0]) : bcastd(row[0], 0);
      if (!(fabs(d) > eps * anorm)) { bad = 1; d = (anorm > 0.0 ? anorm : 1.0); }
      signs |= (d > 0.0) ? 1 : 2;
      double rd = fast_rcp(d);
      if constexpr (DPP) {
        DiagSteps<0>::run(row, d, rd, bad, signs, nb, eps * anorm, anorm, dl, rdl, lane);
      } else {
#pragma unroll
      for (int k = 0; k < LDLR_NB; ++k) {
        // column k of the current block, element j, is A[j][k] = A[k][j]: lane k holds it as row[j] (the strict
        // upper triangle is kept up to date by the same updates), so it reaches all lanes by v_readlane
        const double lik = row[k] * rd;
        double dn = 1.0, rdn = 1.0;
        if (k + 1 < LDLR_NB) {
          row[k + 1] -= lik * bcastd(row[k + 1], k);
          dn = bcastd(row[k + 1], k + 1);
          if (k + 1 < nb) {
            if (!(fabs(dn) > eps * anorm)) { bad = 1; dn = (anorm > 0.0 ? anorm : 1.0); }
            signs |= (dn > 0.0) ? 1 : 2;
          } else {
            dn = 1.0;
          }
          rdn = fast_rcp(dn);
        }
#pragma unroll
        for (int j = k + 2; j < LDLR_NB; ++j) row[j] -= lik * bcastd(row[j], k);
        row[k] = lik;
        if (lane == 0) { dl[k] = d; rdl[k] = rd; }
        d = dn; rd = rdn;
      }
      }
      if (lane < nb) {
#pragma unroll
        for (int j = 0; j < LDLR_NB; ++j) P[lane][j] = row[j];
      }
      if (lane == 0) { if (bad) sflags[0] = 1; sflags[1] |= signs; }
    }
    lds_barrier();
    // finished diagonal block -> global (unit lower L11, pivots on the diagonal)
    if (tid < nb * nb) {
      const int i = tid % nb, j = tid / nb;
      if (i > j) A[(j0 + i) + (size_t)(j0 + j) * lda] = P[i][j];
      else if (i == j) { A[(j0 + i) + (size_t)(j0 + j) * lda] = dl[i]; dvec[j0 + i] = dl[i]; }
    }
    // (c) panel: W = A21 L11^{-T} (thread = row), L21 = W D^{-1} -> LDS and global
    if (64 * wv_here < m) {   // (whole waves: the L11 broadcasts below are wave-wide)
      const int r = nb + min(tid, m - 1);
      // L11 stays in registers, lane k (mod 16) holding its row k; an element reaches the row solves by v_readlane
      // (no LDS round trip inside the dependent chain of a row solve)
      double lrow[LDLR_NB], wrow[LDLR_NB];
#pragma unroll
      for (int k = 0; k < LDLR_NB; ++k) { lrow[k] = P[lane & 15][k]; wrow[k] = P[r][k]; }
      // right-looking order: the updates of one step are independent of each other, only 16 steps are chained
      if constexpr (DPP) {
        PanelSolveCols<0>::run(wrow, lrow);
      } else {
#pragma unroll
        for (int j = 0; j + 1 < LDLR_NB; ++j) {
#pragma unroll
          for (int k = j + 1; k < LDLR_NB; ++k) wrow[k] -= wrow[j] * bcastd_after(lrow[j], k, wrow[j]);
        }
      }
      if (tid < m) {
#pragma unroll
        for (int k = 0; k < LDLR_NB; ++k) {
          if (k < nb) {
            const double l = wrow[k] * rdl[k];
            P[r][k] = l;
            A[(j0 + r) + (size_t)(j0 + k) * lda] = l;
          }
        }
      }
    }
    lds_barrier();
    // (d) trailing update of the tiles still owned: A22 -= (L21 D) L21^T, operands from LDS
    if (m > 0) {
#pragma unroll
      for (int s = 0; s < LDLR_TPW; ++s) {
        const int tI = tIJ[s] >> 8, tJ = tIJ[s] & 255;
        if (tIJ[s] >= 0 && tJ > jt) {
          const int ra = 16 * (tI - jt) + liv, rb = 16 * (tJ - jt) + liv;
          double av[LDLR_NB / 4], bv[LDLR_NB / 4];
#pragma unroll
          for (int q = 0; q < LDLR_NB / 4; ++q) {
            const int k = 4 * q + lkv;
            av[q] = -P[ra][k] * dl[k];
            bv[q] = P[rb][k];
          }
#pragma unroll
          for (int q = 0; q < LDLR_NB / 4; ++q) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc[s], 0, 0, 0);
        }
      }
    }
    lds_barrier();
  }
  if (tid == 0) {
    if (BLOCK) {
      if (sflags[0]) atomicOr(&gflags[0], 1);
      atomicOr(&gflags[1], sflags[1]);
    } else {
      const bool ok = (sflags[0] == 0) && (sflags[1] == 1 || sflags[1] == 2 || n == 0);
      mode[0] = ok ? 1 : 0;
      if (ok) { info[0] = (sflags[1] == 1) ? n : 0; info[1] = (sflags[1] == 2) ? n : 0; info[2] = 0; }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Large S (n > 512: the 1000 x 1000 of configuration C5), round 4: right-looking blocked LDL^T with fat panels of
// DNP = 208 columns (13 tiles: what k_ldl_regs holds in the accumulators of one workgroup).  Per panel three launches:
//   k_ldl_regs<., BLOCK>   the diagonal block, in place (one workgroup: the serial chain of the factorisation)
//   k_dn_trsm              L21 = A21 L11^{-T} D^{-1}: 64 rows per workgroup, blocked forward substitution over the 13
//                          column blocks on the matrix cores (explicit inverses of the 16 x 16 unit diagonal blocks)
//   k_dn_update            A22 -= (L21 D) L21^T, one 16 x 16 tile per wave, K = 208
// (round 3: 32-column panels, two launches each, the 32 x 32 diagonal block factorised redundantly by every workgroup:
// 0.92 ms at n = 1000; the kernels of that form are kept for the measurement switch PP_DENSE_PANEL32.)
constexpr int DNP = 16 * LDLR_NT;       // panel width
constexpr int DNT_LD = 18;              // LDS row stride of a 16-column block (conflict-free operand reads, as LDLR_LD)

// Workgroups beyond the nrb row blocks of A21 run the same substitution on rows of the IDENTITY (64 each): they produce
// inv(L11)^T -- the panel's diagonal block inverted, which turns the coupling solve into matrix-vector products
// (k_dn_fwd / k_dn_bwd) -- in both orientations: Zf[i + k ldz] = Zb[k + i ldz] = inv(L11)[i][k].  The panel is always 208
// columns wide here (run-time masks for a narrower one turned the unrolled substitution into 189 branches and 400 bytes of
// scratch per lane: 36 -> 140 us); the last, narrower panel of a matrix is inverted from a copy padded with the identity
// (k_dn_pad), through this same kernel.
__global__ __launch_bounds__(256) void k_dn_trsm(int n, double* __restrict__ A, const double* __restrict__ dvec, int J0, int nb,
                                                 int nrb, double* __restrict__ Zf, double* __restrict__ Zb) {
  extern __shared__ double dn_lds[];
  // W[jb][row][k]: the panel rows of this workgroup, block by block; Li[jb][c][k] = inv(L11[jb, jb])[c][k]
  double (*W)[64][DNT_LD] = (double (*)[64][DNT_LD])dn_lds;
  double (*Li)[16][17] = (double (*)[16][17])(dn_lds + (size_t)LDLR_NT * 64 * DNT_LD);
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const size_t lda = (size_t)n;
  const int J1 = J0 + nb, m = n - J1, ntb = (nb + 15) / 16;
  const bool ident = (int)blockIdx.x >= nrb;                // rows of the identity instead of rows of A21
  const int R0 = ((int)blockIdx.x - (ident ? nrb : 0)) * 64;      // first row of this workgroup (relative to J1 / in the identity)
  // (1) the 64 x 208 rows of A21: coalesced column reads -> LDS, 13 requests of a thread in flight (one at a time, as a
  // rolled loop issues them, is 52 dependent round trips)
  {
    const int r = tid & 63, c0 = tid >> 6;
    const bool live = !ident && R0 + r < m;
    const double* src = A + (size_t)(J1 + (live ? R0 + r : 0)) + (size_t)J0 * lda;
    for (int cb = 0; cb < DNP; cb += 52) {
      double v[13];
#pragma unroll
      for (int u = 0; u < 13; ++u) v[u] = ident ? ((R0 + r == cb + c0 + 4 * u) ? 1.0 : 0.0) : src[(size_t)(cb + c0 + 4 * u) * lda];
#pragma unroll
      for (int u = 0; u < 13; ++u) { const int c = cb + c0 + 4 * u; W[c >> 4][r][c & 15] = (live || ident) ? v[u] : 0.0; }
    }
  }
  // (2) inverses of the unit lower 16 x 16 diagonal blocks of L11.  The blocks go to LDS first (one request per thread
  // and block: a chain of 120 global loads per inverse was most of this kernel); then lane c < 16 of wave (jb mod 4) owns
  // column c of inv(L_jj): X[i][c] = delta_ic - sum_{k < i} L[i][k] X[k][c], the entries of L broadcast from LDS
  for (int jb = 0; jb < LDLR_NT; ++jb) {
    const int i = tid & 15, k = tid >> 4;
    Li[jb][i][k] = A[(size_t)(J0 + 16 * jb + i) + (size_t)(J0 + 16 * jb + k) * lda];
  }
  __syncthreads();
  for (int jb = wv; jb < LDLR_NT; jb += 4) {
    double x[16];
    if (lane < 16) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        double v = (i == lane) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) v -= Li[jb][i][k] * x[k];
        x[i] = v;
      }
    }
    __builtin_amdgcn_wave_barrier();           // (the block is read by this wave only: overwrite it with its inverse)
    if (lane < 16) {
#pragma unroll
      for (int i = 0; i < 16; ++i) Li[jb][i][lane] = x[i];       // Li[jb][row i][column c]
    }
  }
  __syncthreads();
  // (3) blocked forward substitution; every wave owns 16 rows and never reads another wave's.  The loop over the block
  // rows is unrolled completely (a panel with rows below it always has LDLR_NT = 13 of them): every L11 operand address
  // is then a constant offset and the compiler requests the operands of the following block rows (from global memory /
  // L2) while it works on the current one -- with one memory round trip in front of every block row the kernel took 86 us.
  const int wr = 16 * wv;
  const double* Lrow0 = A + (size_t)(J0 + li) + (size_t)(J0 + lk) * lda;      // L11[n = li][k = lk]
  constexpr int KBMAX = LDLR_NT - 1;
  // operands of block rows jb + 1 and jb + 2, requested two block rows ahead of their use (B[k][n] = L[16 jb + n][16 kb + k])
  double b1[KBMAX][4], b2[KBMAX][4];
#define PP_TRSM_LOAD(dst, JB)                                                                            \
  _Pragma("unroll") for (int kb_ = 0; kb_ < KBMAX; ++kb_)                                                \
    if (kb_ < (JB) && (JB) < LDLR_NT)                                                                    \
      _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_)                                                   \
        dst[kb_][q_] = Lrow0[(size_t)(16 * (JB)) + (size_t)(16 * kb_ + 4 * q_) * lda];
  PP_TRSM_LOAD(b1, 1)
  PP_TRSM_LOAD(b2, 2)
#pragma unroll
  for (int jb = 0; jb < LDLR_NT; ++jb) {
    double4_t acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = W[jb][wr + lk + 4 * r][li];
    double bc[KBMAX][4];
    if (jb >= 1) {
#pragma unroll
      for (int kb = 0; kb < KBMAX; ++kb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { bc[kb][q] = b1[kb][q]; b1[kb][q] = b2[kb][q]; }
      }
      PP_TRSM_LOAD(b2, jb + 2)
    }
    // acc -= sum_kb W_kb (16 x 16) L[jb, kb]^T; four accumulators: a chain of up to 48 dependent matrix instructions on
    // one accumulator was 23 of the kernel's 43 us (timing builds without the products / without the inverses)
    double4_t part[3] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
#pragma unroll
    for (int kb = 0; kb < jb; ++kb) {
      double av[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) av[q] = -W[kb][wr + li][4 * q + lk];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if ((kb & 3) == 0) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bc[kb][q], acc, 0, 0, 0);
        else part[(kb & 3) - 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bc[kb][q], part[(kb & 3) - 1], 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += (part[0][r] + part[1][r]) + part[2][r];
    // W_jb = acc inv(L_jj)^T: through LDS into operand layout (own rows only: wave-local ordering suffices)
#pragma unroll
    for (int r = 0; r < 4; ++r) W[jb][wr + lk + 4 * r][li] = acc[r];
    // (LDS operations of one wave execute in order: its own writes are visible to its reads without a wait)
    double4_t w = {0.0, 0.0, 0.0, 0.0};
    {
      double av[4], bv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { av[q] = W[jb][wr + li][4 * q + lk]; bv[q] = Li[jb][li][4 * q + lk]; }   // B[k][n] = inv(L)[n][k]
#pragma unroll
      for (int q = 0; q < 4; ++q) w = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], w, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) W[jb][wr + lk + 4 * r][li] = w[r];
  }
#undef PP_TRSM_LOAD
  __syncthreads();
  // (4) identity rows: inv(L11)^T in both orientations;  rows of A21: L21 = W D^{-1}, coalesced column writes
  if (ident) {
    // W[.][r][c] = inv(L11)^T [i = R0 + r][c] = inv(L11)[c][i]; both orientations written with the fast index across the
    // threads (scattered 8-byte stores made this kernel 140 us)
    {
      const int r = tid & 63, c0 = tid >> 6, i = R0 + r;
      if (i < nb)
        for (int c = c0; c < nb; c += 4) Zb[(size_t)i + (size_t)c * DNP] = W[c >> 4][r][c & 15];
    }
    if (tid < nb) {
      const int c = tid;
      for (int r = 0; r < 64 && R0 + r < nb; ++r) Zf[(size_t)c + (size_t)(R0 + r) * DNP] = W[c >> 4][r][c & 15];
    }
  } else {
    const int r = tid & 63, c0 = tid >> 6;
    if (R0 + r < m) {
      double* dst = A + (size_t)(J1 + R0 + r) + (size_t)J0 * lda;
#pragma unroll 13
      for (int c = c0; c < DNP; c += 4) dst[(size_t)c * lda] = W[c >> 4][r][c & 15] / dvec[J0 + c];
    }
  }
}

// Coupling solve with the fat-panel factor (n > 512, accepted unpivoted factorisation): x = L^-T D^-1 L^-1 b panel by panel
// with the inverted diagonal blocks of k_dn_trsm -- every step a matrix-vector product across the chip instead of a
// triangular solve in one workgroup (one workgroup for the whole 1000 x 1000 factor: 0.26 ms).
//   k_dn_fwd   every workgroup forms y_p = inv(L_pp) b_p (redundantly: 208 x 208) and subtracts L[r, p] y_p from its 256
//              rows below the panel; workgroup 0 stores y_p.  first: b = r_c + r_s is read instead of the work vector.
//   k_dn_bwdA  partial sums of L[rows, p]^T x over 64-row chunks (fixed layout: deterministic)
//   k_dn_bwdB  x_p = inv(L_pp)^T (y_p / d - sum of the partials)
// The nb x nb unit lower diagonal block of the last panel, padded with the identity to 208 x 208 (leading dimension 208)
__global__ __launch_bounds__(256) void k_dn_pad(int n, const double* __restrict__ A, int J0, int nb, double* __restrict__ P) {
  const size_t lda = (size_t)n;
  for (int idx = (int)blockIdx.x * 256 + threadIdx.x; idx < DNP * DNP; idx += (int)gridDim.x * 256) {
    const int i = idx % DNP, k = idx / DNP;
    P[idx] = (i < nb && k < nb) ? ((i > k) ? A[(size_t)(J0 + i) + (size_t)(J0 + k) * lda] : (i == k ? 1.0 : 0.0)) : (i == k ? 1.0 : 0.0);
  }
}

constexpr int DNS_THREADS = 4 * DNP;      // 832 = 13 waves: 208 outputs x 4 slices of the 208 terms (52 requests per thread, one round trip)
constexpr int DNS_Q = DNP / 4;

__global__ __launch_bounds__(DNS_THREADS) void k_dn_fwd(int n, const double* __restrict__ A, const double* __restrict__ Zf, int J0, int nb,
                                                        const double* __restrict__ rc, const double* __restrict__ rs, int first,
                                                        double* __restrict__ b, double* __restrict__ y, const int* __restrict__ mode) {
  __shared__ double bp[DNP], yp[DNP], red[4][DNP];
  if (mode[0] != 1) return;
  const int tid = threadIdx.x, i = tid % DNP, q = tid / DNP;
  const size_t lda = (size_t)n;
  const int J1 = J0 + nb;
  if (tid < DNP) bp[tid] = tid < nb ? (first ? ((rc ? rc[J0 + tid] : 0.0) + rs[J0 + tid]) : b[J0 + tid]) : 0.0;
  __syncthreads();
  {
    // y_p = inv(L_pp) b_p: thread (i, q) takes the terms k = 52 q .. 52 q + 51 of row i (entries beyond nb are not defined)
    const double* Zr = Zf + min(i, nb - 1);
    double z[DNS_Q];
#pragma unroll
    for (int u = 0; u < DNS_Q; ++u) z[u] = Zr[(size_t)min(DNS_Q * q + u, nb - 1) * DNP];
    double sum = 0.0;
#pragma unroll
    for (int u = 0; u < DNS_Q; ++u) sum += z[u] * bp[DNS_Q * q + u];          // (bp is zero beyond nb; no select: it would become a branch per term)
    red[q][i] = sum;
  }
  __syncthreads();
  if (tid < DNP) {
    const double v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    yp[tid] = tid < nb ? v : 0.0;
    if (blockIdx.x == 0 && tid < nb) y[J0 + tid] = v;
  }
  __syncthreads();
  if (blockIdx.x == 0) return;
  // rows below the panel: 208 per workgroup, the 208 terms of a row in 4 slices
  const int r = J1 + ((int)blockIdx.x - 1) * DNP + i;
  const bool live = r < n;
  {
    const double* Lr = A + (size_t)(live ? r : J1) + (size_t)J0 * lda;
    double l[DNS_Q];
#pragma unroll
    for (int u = 0; u < DNS_Q; ++u) l[u] = Lr[(size_t)min(DNS_Q * q + u, nb - 1) * lda];
    double sum = 0.0;
#pragma unroll
    for (int u = 0; u < DNS_Q; ++u) sum += l[u] * yp[DNS_Q * q + u];          // (yp is zero beyond nb)
    red[q][i] = sum;
  }
  __syncthreads();
  if (tid < DNP && live) {
    const double acc = first ? ((rc ? rc[r] : 0.0) + rs[r]) : b[r];
    b[r] = acc - ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
  }
}

__global__ __launch_bounds__(DNS_THREADS) void k_dn_bwdA(int n, const double* __restrict__ A, int J0, int nb, const double* __restrict__ x,
                                                         double* __restrict__ part, const int* __restrict__ mode) {
  __shared__ double xr[64], red[4][DNP];
  if (mode[0] != 1) return;
  const int tid = threadIdx.x, k = tid % DNP, q = tid / DNP;
  const size_t lda = (size_t)n;
  const int r0 = J0 + nb + (int)blockIdx.x * 64, nr = min(64, n - r0);
  if (tid < 64) xr[tid] = tid < nr ? x[r0 + tid] : 0.0;
  __syncthreads();
  {
    const double* Lc = A + (size_t)r0 + (size_t)(J0 + min(k, nb - 1)) * lda;
    double l[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) l[u] = Lc[min(16 * q + u, nr - 1)];
    double sum = 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) sum += l[u] * xr[16 * q + u];                 // (xr is zero beyond nr)
    red[q][k] = sum;
  }
  __syncthreads();
  if (tid < nb) part[(size_t)blockIdx.x * DNP + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

__global__ __launch_bounds__(DNS_THREADS) void k_dn_bwdB(int n, const double* __restrict__ Zb, const double* __restrict__ dvec, int J0, int nb,
                                                         int nchunk, const double* __restrict__ y, const double* __restrict__ part,
                                                         double* __restrict__ x, const int* __restrict__ mode) {
  __shared__ double vp[DNP], red[4][DNP];
  if (mode[0] != 1) return;
  const int tid = threadIdx.x, i = tid % DNP, q = tid / DNP;
  if (tid < DNP) {
    double t = 0.0;
    if (tid < nb) {
      double pv[16];
#pragma unroll
      for (int g = 0; g < 16; ++g) pv[g] = g < nchunk ? part[(size_t)g * DNP + tid] : 0.0;       // (at most 13 chunks of 64 rows below a panel)
#pragma unroll
      for (int g = 0; g < 16; ++g) t += pv[g];
    }
    vp[tid] = tid < nb ? y[J0 + tid] / dvec[J0 + tid] - t : 0.0;
  }
  __syncthreads();
  {
    // x_i = sum_k inv(L_pp)[k][i] v_k (k >= i; the entries with k < i are exact zeros)
    const double* Zr = Zb + min(i, nb - 1);
    double z[DNS_Q];
#pragma unroll
    for (int u = 0; u < DNS_Q; ++u) z[u] = Zr[(size_t)min(DNS_Q * q + u, nb - 1) * DNP];
    double sum = 0.0;
#pragma unroll
    for (int u = 0; u < DNS_Q; ++u) sum += z[u] * vp[DNS_Q * q + u];          // (vp is zero beyond nb)
    red[q][i] = sum;
  }
  __syncthreads();
  if (tid < nb) x[J0 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

__global__ __launch_bounds__(DN_THREADS) void k_dn_update(int n, double* __restrict__ A, const double* __restrict__ dvec, int J0,
                                                          int nb) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t lda = (size_t)n;
  const int J1 = J0 + nb, m = n - J1;
  const int nt = (m + 15) / 16, ntiles = nt * (nt + 1) / 2;
  const int tix = (int)blockIdx.x * (DN_THREADS / 64) + wv;
  if (tix >= ntiles) return;
  const int li = lane & 15, lk = lane >> 4;
  int I = (int)((sqrt(8.0 * tix + 1.0) - 1.0) * 0.5);
  while ((I + 1) * (I + 2) / 2 <= tix) ++I;
  while (I * (I + 1) / 2 > tix) --I;
  const int J = tix - I * (I + 1) / 2;
  const int ra = J1 + 16 * I + li, rb = J1 + 16 * J + li;
  const bool va = ra < n, vb = rb < n;
  const double* pa = A + (size_t)(va ? ra : J1) + (size_t)(J0 + lk) * lda;
  const double* pb = A + (size_t)(vb ? rb : J1) + (size_t)(J0 + lk) * lda;
  const double* pd = dvec + J0 + lk;
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  constexpr int QB = 26;                             // K steps requested together (two rounds cover a 208-column panel)
  for (int q0 = 0; q0 < nb / 4; q0 += QB) {          // (nb is a multiple of 16 for every panel that has a trailing matrix)
    double av[QB], bv[QB], dv[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const bool in = q0 + q < nb / 4;
      const size_t off = (size_t)(4 * (in ? q0 + q : 0)) * lda;
      av[q] = (in && va) ? pa[off] : 0.0;
      bv[q] = (in && vb) ? pb[off] : 0.0;
      dv[q] = in ? pd[4 * (q0 + q)] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q] * dv[q], bv[q], acc, 0, 0, 0);
  }
  const int col = J1 + 16 * J + li;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = J1 + 16 * I + lk + 4 * r;
    if (row < n && col < n && row >= col) A[row + (size_t)col * lda] -= acc[r];
  }
}

// x = S^-1 b with the blocked factor (unit lower L in A, pivots in dvec); b is in LDS vector x
__device__ void ldl_blocked_solve(int n, const double* __restrict__ A, const double* __restrict__ dvec, double* x) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nwv = blockDim.x >> 6;
  const size_t lda = (size_t)n;
  for (int j0 = 0; j0 < n; j0 += LDL_NB) {
    const int nb = min(LDL_NB, n - j0), j1 = j0 + nb;
    if (wv == 0) {  // unit-lower triangular solve inside the block: lane = row, shuffle broadcast
      double lrow[LDL_NB];
#pragma unroll
      for (int k = 0; k < LDL_NB; ++k)
        lrow[k] = (lane > k && lane < nb) ? A[(j0 + lane) + (size_t)(j0 + k) * lda] : 0.0;
      double xi = (lane < nb) ? x[j0 + lane] : 0.0;
#pragma unroll
      for (int k = 0; k < LDL_NB; ++k) xi -= lrow[k] * bcastd(xi, k);
      if (lane < nb) x[j0 + lane] = xi;
    }
    __syncthreads();
    for (int r = j1 + tid; r < n; r += blockDim.x) {
      double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
      for (int k = 0; k < LDL_NB; k += 2) {
        if (k < nb) s0 += A[r + (size_t)(j0 + k) * lda] * x[j0 + k];
        if (k + 1 < nb) s1 += A[r + (size_t)(j0 + k + 1) * lda] * x[j0 + k + 1];
      }
      x[r] -= s0 + s1;
    }
    __syncthreads();
  }
  for (int i = tid; i < n; i += blockDim.x) x[i] /= dvec[i];
  __syncthreads();
  for (int j0 = ((n - 1) / LDL_NB) * LDL_NB; j0 >= 0; j0 -= LDL_NB) {
    const int nb = min(LDL_NB, n - j0), j1 = j0 + nb;
    for (int k = wv; k < nb; k += nwv) {  // x[j0+k] -= L[j1:, j0+k]^T x[j1:]
      double s = 0.0;
      for (int r = j1 + lane; r < n; r += 64) s += A[r + (size_t)(j0 + k) * lda] * x[r];
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
      if (lane == 0) x[j0 + k] -= s;
    }
    __syncthreads();
    if (wv == 0) {
      double lcol[LDL_NB];
#pragma unroll
      for (int k = 0; k < LDL_NB; ++k)
        lcol[k] = (lane < k && k < nb) ? A[(j0 + k) + (size_t)(j0 + lane) * lda] : 0.0;
      double xi = (lane < nb) ? x[j0 + lane] : 0.0;
#pragma unroll
      for (int k = LDL_NB - 1; k > 0; --k) xi -= lcol[k] * bcastd(xi, k);
      if (lane < nb) x[j0 + lane] = xi;
    }
    __syncthreads();
  }
}

// Same solve for n <= blockDim.x with one thread per unknown and ONE barrier per NBS-column block: thread r
// keeps its right-hand-side entry in a register and its NBS-entry segment of the current block of L
// (prefetched one block ahead, so no global-memory round trip sits between the dependent blocks).  The
// wave that owns the block's rows solves it with v_readlane broadcasts, which at the same time applies the
// block to the other rows of that wave; the remaining waves apply it from LDS after the barrier.
template <int NBS>
__device__ void ldl_rows_solve(int n, const double* __restrict__ A, const double* __restrict__ dvec, double* xs) {
  const int r = threadIdx.x, lane = r & 63, wv = __builtin_amdgcn_readfirstlane(r >> 6);
  const size_t lda = (size_t)n;
  const bool act = r < n;
  double acc = act ? xs[r] : 0.0;
  const double rd = act ? 1.0 / dvec[r] : 0.0;
  // three segment buffers, rotated by unrolling the block loop three times: the segment of block j + 2 is requested
  // before block j is solved, so a load has two block steps (not the rest of one) to arrive
  double c0[NBS], c1[NBS], c2[NBS];
  (void)lane;
  // ---- forward: L y = b, blocks ascending; segment = L[r][j0 .. j0+NBS) below the diagonal
  // A wave whose 64 rows all lie below the block needs no per-element predicate (and one whose rows all lie above it
  // loads nothing): the predicated form costs ~12 instructions per element, which for 8 waves x 32 elements was most
  // of a block step.  Only the wave that holds the block's own rows takes the predicated path.
  const int row_lo = 64 * wv, row_hi = 64 * wv + 63;
#define PP_LOAD_FWD(dst, j0_)                                                              \
  {                                                                                        \
    const int jl = (j0_);                                                                  \
    if (row_lo >= jl + NBS && row_hi < n && jl + NBS <= n) {                               \
      const double* src = A + r + (size_t)jl * lda;                                        \
      _Pragma("unroll") for (int k = 0; k < NBS; ++k) dst[k] = src[(size_t)k * lda];       \
    } else if (row_hi < jl || jl >= n) {                                                   \
      _Pragma("unroll") for (int k = 0; k < NBS; ++k) dst[k] = 0.0;                        \
    } else {                                                                               \
      _Pragma("unroll") for (int k = 0; k < NBS; ++k) {                                    \
        const int c = jl + k;                                                              \
        dst[k] = (act && c < n && c < r) ? A[r + (size_t)c * lda] : 0.0;                   \
      }                                                                                    \
    }                                                                                      \
  }
#define PP_STEP_FWD(cur, j0_)                                                              \
  {                                                                                        \
    const int jj = (j0_), bw = jj >> 6, base = jj & 63;                                    \
    if (wv == bw) {                                                                        \
      _Pragma("unroll") for (int k = 0; k < NBS; ++k) acc -= cur[k] * bcastd(acc, base + k); \
      if (r >= jj && r < jj + NBS && act) xs[r] = acc;                                     \
    }                                                                                      \
    lds_barrier();   /* not __syncthreads(): that would also drain the prefetched global loads */ \
    if (wv != bw && r >= jj + NBS) {                                                       \
      _Pragma("unroll") for (int k = 0; k < NBS; ++k) acc -= cur[k] * xs[min(jj + k, n - 1)]; \
    }                                                                                      \
  }
  PP_LOAD_FWD(c0, 0)
  PP_LOAD_FWD(c1, NBS)
  for (int j0 = 0; j0 < n; j0 += 3 * NBS) {
    PP_LOAD_FWD(c2, j0 + 2 * NBS)
    PP_STEP_FWD(c0, j0)
    if (j0 + NBS >= n) break;
    PP_LOAD_FWD(c0, j0 + 3 * NBS)
    PP_STEP_FWD(c1, j0 + NBS)
    if (j0 + 2 * NBS >= n) break;
    PP_LOAD_FWD(c1, j0 + 4 * NBS)
    PP_STEP_FWD(c2, j0 + 2 * NBS)
  }
#undef PP_LOAD_FWD
#undef PP_STEP_FWD
  acc *= rd;
  // ---- backward: L^T x = y, blocks descending; segment = L[j0 .. j0+NBS)[r] below the diagonal
#define PP_LOAD_BWD(dst, j0_)                                                              \
  {                                                                                        \
    const int jl = (j0_);                                                                  \
    if (jl >= 0 && row_hi < jl && jl + NBS <= n) {                                         \
      const double* src = A + jl + (size_t)r * lda;                                        \
      _Pragma("unroll") for (int k = 0; k < NBS; ++k) dst[k] = src[k];                     \
    } else if (jl < 0 || row_lo >= jl + NBS) {                                             \
      _Pragma("unroll") for (int k = 0; k < NBS; ++k) dst[k] = 0.0;                        \
    } else {                                                                               \
      _Pragma("unroll") for (int k = 0; k < NBS; ++k) {                                    \
        const int c = jl + k;                                                              \
        dst[k] = (act && c >= 0 && c < n && c > r) ? A[c + (size_t)r * lda] : 0.0;         \
      }                                                                                    \
    }                                                                                      \
  }
#define PP_STEP_BWD(cur, j0_)                                                              \
  {                                                                                        \
    const int jj = (j0_), bw = jj >> 6, base = jj & 63;                                    \
    if (wv == bw) {                                                                        \
      _Pragma("unroll") for (int k = NBS - 1; k >= 0; --k) acc -= cur[k] * bcastd(acc, base + k); \
      if (r >= jj && r < jj + NBS && act) xs[r] = acc;                                     \
    }                                                                                      \
    lds_barrier();   /* not __syncthreads(): that would also drain the prefetched global loads */ \
    if (wv != bw && r < jj) {                                                              \
      _Pragma("unroll") for (int k = 0; k < NBS; ++k) acc -= cur[k] * xs[min(jj + k, n - 1)]; \
    }                                                                                      \
  }
  const int jlast = ((n - 1) / NBS) * NBS;
  __syncthreads();
  PP_LOAD_BWD(c0, jlast)
  PP_LOAD_BWD(c1, jlast - NBS)
  for (int j0 = jlast; j0 >= 0; j0 -= 3 * NBS) {
    PP_LOAD_BWD(c2, j0 - 2 * NBS)
    PP_STEP_BWD(c0, j0)
    if (j0 - NBS < 0) break;
    PP_LOAD_BWD(c0, j0 - 3 * NBS)
    PP_STEP_BWD(c1, j0 - NBS)
    if (j0 - 2 * NBS < 0) break;
    PP_LOAD_BWD(c1, j0 - 4 * NBS)
    PP_STEP_BWD(c2, j0 - 2 * NBS)
  }
#undef PP_LOAD_BWD
#undef PP_STEP_BWD
  __syncthreads();
}

// Last kernel of the dense phase.  If the unpivoted factorisation was accepted (mode[0] == 1) it only publishes the
// status; otherwise it builds S + Q in A (Q lower triangle authoritative, may be null) and runs Bunch-Kaufman.
__global__ __launch_bounds__(BK_THREADS) void k_bk_factor(int n, const double* __restrict__ S, const double* __restrict__ Q,
                                                          double* A, int* ipiv, double* work, int* info, const int* mode,
                                                          long long* status_out, long long seq) {
  __shared__ double sv[16];
  __shared__ int si[16];
  if (mode[0] != 1) {
    for (size_t idx = threadIdx.x; idx < (size_t)n * n; idx += BK_THREADS) {
      const int i = (int)(idx % n), j = (int)(idx / n);
      double q = 0.0;
      if (Q) q = (i >= j) ? Q[(size_t)i + (size_t)j * n] : Q[(size_t)j + (size_t)i * n];
      A[idx] = S[idx] + q;
    }
    __syncthreads();
    TeamCtx ctx{sv, si};
    __shared__ pp::BkInfo sbi;
    pp::bk_factor(ctx, n, A, n, ipiv, work, &sbi, BK_EPS);
    if (threadIdx.x == 0) { info[0] = sbi.npos; info[1] = sbi.nneg; info[2] = sbi.nzero; }
  }
  if (threadIdx.x == 0) publish_status(S + (size_t)n * n, info, status_out, seq);
}

// xc = S^-1 (rc + rs): blocked LDL^T factor if it was accepted, else the Bunch-Kaufman factor
// THREADS x NBS: 512 threads with 32-column segments (n_c <= 512), or 1024 threads with 16-column segments
// (512 < n_c <= 1024: two 16-entry segments are what 128 VGPRs per thread leave room for; the global-memory
// fallback ldl_blocked_solve puts three dependent load round trips into each of its 2 n_c / 32 block steps:
// 0.89 ms at n_c = 1000)
template <int THREADS, int NBS>
__global__ __launch_bounds__(THREADS) void k_coupling_solve(int n, const double* Abk, const int* ipiv,
                                                            const double* Aldl, const double* dvec, const int* mode,
                                                            const double* rc, const double* rs, double* xc, int ldl_done = 0) {
  extern __shared__ __attribute__((aligned(16))) double xs[];
  __shared__ double sv[16];
  __shared__ int si[16];
  if (mode[0] == 1 && ldl_done) return;         // (solved by the panel kernels, k_dn_fwd / k_dn_bwd)
  if (mode[0] == 1) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) xs[i] = (rc ? rc[i] : 0.0) + rs[i];
    __syncthreads();
    if (n <= THREADS) ldl_rows_solve<NBS>(n, Aldl, dvec, xs);
    else ldl_blocked_solve(n, Aldl, dvec, xs);
    for (int i = threadIdx.x; i < n; i += blockDim.x) xc[i] = xs[i];
    return;
  }
  TeamCtx ctx{sv, si};
  for (int i = threadIdx.x; i < n; i += blockDim.x) xc[i] = (rc ? rc[i] : 0.0) + rs[i];
  __syncthreads();
  pp::bk_solve(ctx, n, Abk, n, ipiv, xc);
}


// layout of pp_solver::dn_z: [Zf | Zb] per panel, work vectors b and y (n each), partial sums (16 x 208), padded block
size_t dn_pad_offset(int nc) {
  const size_t np = (size_t)(nc + DNP - 1) / DNP;
  return 2 * np * DNP * DNP + 2 * (size_t)nc + 16 * DNP;
}

}  // namespace

int ppi_dense_factor_schur(pp_handle h, const double* Q_host) {
  hipStream_t st = h->stream;
  const int nc = h->nc;
  const size_t nn = schur_doubles(h);
  const bool follow = h->schur_on_side && h->dense_stream && !Q_host && !h->profile;      // the Schur update ran on the dense stream
  h->schur_on_side = false;
  if (!follow) { if (int rc = join_dense(h)) return rc; }
  // The dense factorisation leaves most of the chip idle (one workgroup for n_c <= 512, a chain of small launches
  // beyond): it runs on a stream of its own, forked here, so that a forward sweep enqueued behind this call (it does not
  // depend on S) overlaps it; the coupling solve joins.
  // Not with a host Q (its upload is ordered by the caller on the handle's stream) and not while phases are timed.
  // (nor with more than two pattern groups: their streams and this one would share hardware queues)
  const bool overlap = h->dense_overlap && !h->profile && !Q_host && h->dense_policy == 0 && h->groups.size() <= 2;
  if (follow) {
    h->dense_pending = false;       // (re-armed below: the handle's stream joins behind the whole dense phase)
    st = h->dense_stream;
  } else if (overlap) {
    if (!h->dense_stream) {
      PP_HIP(hipStreamCreateWithFlags(&h->dense_stream, hipStreamNonBlocking));
      PP_HIP(hipEventCreateWithFlags(&h->ev_dense_fork, hipEventDisableTiming));
      PP_HIP(hipEventCreateWithFlags(&h->ev_dense_done, hipEventDisableTiming));
    }
    PP_HIP(hipEventRecord(h->ev_dense_fork, h->stream));
    PP_HIP(hipStreamWaitEvent(h->dense_stream, h->ev_dense_fork, 0));
    st = h->dense_stream;
  }
  {
    if (Q_host) PP_HIP(hipMemcpyAsync(h->Qd, Q_host, nn * sizeof(double), hipMemcpyHostToDevice, st));
    h->have_Q = Q_host != nullptr;      // (the a-posteriori check of the back-solves reads it: refine.hip)
    const double* Qd = Q_host ? h->Qd : nullptr;
    const bool regs = h->dense_policy == 0 && nc <= 16 * LDLR_NT;
    PhaseScope ps(h, 3, regs ? 2 : 3);
    // (the register-resident kernel reads S + Q itself; the global-memory variants work in place on a copy)
    if (h->dense_policy == 0 && !regs)
      hipLaunchKernelGGL(k_add_q, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, h->S, Qd, h->Sfac, h->Sldl, nc);
    if (h->dense_policy == 0)
      // (a left-looking variant with the panel resident in LDS was measured no faster: 0.344 vs 0.315 ms at
      // n_c = 200 -- the serial diagonal-block factor dominates both)
      if (nc <= 16 * LDLR_NT) {
        if (h->dense_dpp) hipLaunchKernelGGL(k_ldl_regs<true>, dim3(1), dim3(LDL_THREADS), 0, st, nc, h->S, Qd, h->Sldl, h->dvec, h->dense_mode, h->bkinfo,
                           BK_EPS);
        else hipLaunchKernelGGL(k_ldl_regs<false>, dim3(1), dim3(LDL_THREADS), 0, st, nc, h->S, Qd, h->Sldl, h->dvec, h->dense_mode, h->bkinfo,
                           BK_EPS);
      } else if (nc <= 512) {
        hipLaunchKernelGGL(k_ldl_blocked, dim3(1), dim3(LDL_THREADS), 0, st, nc, h->Sldl, h->dvec, h->dense_mode, h->bkinfo,
                           BK_EPS);
      } else {
        // large S: fat panels of 208 columns (diagonal block by the register-resident kernel, panel solve and trailing
        // update on the matrix cores across the chip); PP_DENSE_PANEL32: the 32-column form of round 3
        double* anorm = h->work;                    // (scratch of the Bunch-Kaufman fallback, free until then:
        double* stage = h->work + 8;                //  2 n_c doubles >= 8 + 32 * 32 for n_c > 512)
        int* flags = h->dense_mode + 2;
        hipLaunchKernelGGL(k_dense_anorm, dim3(1), dim3(256), 0, st, nc, h->Sldl, anorm, flags);
        static const bool panel32 = pp::env_switch("PP_DENSE_PANEL32") != nullptr;
        if (panel32) {
          for (int j0 = 0; j0 < nc; j0 += LDL_NB) {
            const int m = nc - std::min(nc, j0 + LDL_NB);
            hipLaunchKernelGGL(k_dense_panel, dim3(1 + (m + DN_THREADS - 1) / DN_THREADS), dim3(DN_THREADS), 0, st, nc,
                               h->Sldl, h->dvec, anorm, flags, stage, j0, BK_EPS);
            if (m > 0) {
              const int nt = (m + 15) / 16, ntiles = nt * (nt + 1) / 2, per = DN_THREADS / 64;
              hipLaunchKernelGGL(k_dense_update, dim3((ntiles + per - 1) / per), dim3(DN_THREADS), 0, st, nc, h->Sldl,
                                 h->dvec, stage, j0);
            }
          }
        } else {
          const size_t trsm_lds = ((size_t)LDLR_NT * 64 * DNT_LD + (size_t)LDLR_NT * 16 * 17) * sizeof(double);
          if (!h->dn_z) {       // inverted diagonal blocks (two orientations per panel) + work vectors of the panel solve
            const size_t np = (size_t)(nc + DNP - 1) / DNP;
            void* zp = nullptr;
            if (hipMalloc(&zp, (dn_pad_offset(nc) + (size_t)DNP * DNP) * sizeof(double)) != hipSuccess)
              return fail(h, 1, "hipMalloc failed (dense panel inverses)");
            h->dn_z = (double*)zp;
          }
          if (!h->dn_lds_attr) {
            PP_HIP(hipFuncSetAttribute((const void*)k_dn_trsm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)trsm_lds));
            h->dn_lds_attr = true;
          }
          for (int j0 = 0; j0 < nc; j0 += DNP) {
            const int nb = std::min(DNP, nc - j0), m = nc - j0 - nb;
            double* blk = h->Sldl + (size_t)j0 + (size_t)j0 * nc;
            if (h->dense_dpp) hipLaunchKernelGGL((k_ldl_regs<true, true>), dim3(1), dim3(LDL_THREADS), 0, st, nb, blk, (const double*)nullptr, blk,
                                                 h->dvec + j0, (int*)nullptr, (int*)nullptr, BK_EPS, nc, anorm, flags);
            else hipLaunchKernelGGL((k_ldl_regs<false, true>), dim3(1), dim3(LDL_THREADS), 0, st, nb, blk, (const double*)nullptr, blk,
                                    h->dvec + j0, (int*)nullptr, (int*)nullptr, BK_EPS, nc, anorm, flags);
            // (the workgroups behind the row blocks of A21 invert the diagonal block for the coupling solve; a last panel
            // narrower than 208 columns from its identity-padded copy)
            double* Zf = h->dn_z + (size_t)2 * (j0 / DNP) * DNP * DNP;
            if (nb == DNP) {
              hipLaunchKernelGGL(k_dn_trsm, dim3((m + 63) / 64 + (DNP + 63) / 64), dim3(256), trsm_lds, st, nc, h->Sldl, h->dvec, j0, nb,
                                 (m + 63) / 64, Zf, Zf + (size_t)DNP * DNP);
            } else {
              double* pad = h->dn_z + dn_pad_offset(nc);
              hipLaunchKernelGGL(k_dn_pad, dim3(32), dim3(256), 0, st, nc, h->Sldl, j0, nb, pad);
              hipLaunchKernelGGL(k_dn_trsm, dim3((DNP + 63) / 64), dim3(256), trsm_lds, st, DNP, pad, h->dvec, 0, DNP, 0, Zf,
                                 Zf + (size_t)DNP * DNP);
            }
            if (m > 0) {
              const int nt = (m + 15) / 16, ntiles = nt * (nt + 1) / 2, per = DN_THREADS / 64;
              hipLaunchKernelGGL(k_dn_update, dim3((ntiles + per - 1) / per), dim3(DN_THREADS), 0, st, nc, h->Sldl, h->dvec, j0, nb);
            }
          }
        }
        hipLaunchKernelGGL(k_dense_finish, dim3(1), dim3(64), 0, st, nc, flags, h->dense_mode, h->bkinfo);
      }
    else
      PP_HIP(hipMemsetAsync(h->dense_mode, 0, sizeof(int), st));
    // Bunch-Kaufman on S + Q if the unpivoted factorisation was not accepted; publishes the status either way
    hipLaunchKernelGGL(k_bk_factor, dim3(1), dim3(BK_THREADS), 0, st, nc, h->S, Qd, h->Sfac, h->ipiv, h->work, h->bkinfo,
                       h->dense_mode, h->status_dev, ++h->status_seq);
  }
  if (overlap || follow) {
    PP_HIP(hipEventRecord(h->ev_dense_done, st));
    h->dense_pending = true;
  }
  return 0;
}

int ppi_dense_coupling_solve(pp_handle h, const double* rc_dev) {
  if (int rc = join_dense(h)) return rc;
  hipStream_t st = h->stream;
  const int nc = h->nc;
  static const bool panel32 = pp::env_switch("PP_DENSE_PANEL32") != nullptr;
  const bool panels = nc > 512 && h->dn_z && h->dense_policy == 0 && !panel32;
  PhaseScope ps(h, 6, panels ? 15 : 1);
  if (panels) {
    // accepted unpivoted factor (decided on the device: the kernels leave at once otherwise and k_coupling_solve does the
    // Bunch-Kaufman solve): forward and backward substitution over the 208-column panels, matrix-vector products only
    const int np = (nc + DNP - 1) / DNP;
    double* b = h->dn_z + (size_t)2 * np * DNP * DNP;
    double* y = b + nc;
    double* part = y + nc;
    for (int p = 0; p < np; ++p) {
      const int j0 = p * DNP, nb = std::min(DNP, nc - j0), m = nc - j0 - nb;
      hipLaunchKernelGGL(k_dn_fwd, dim3(1 + (m + DNP - 1) / DNP), dim3(DNS_THREADS), 0, st, nc, h->Sldl, h->dn_z + (size_t)2 * p * DNP * DNP, j0, nb,
                         rc_dev, h->rs, p == 0 ? 1 : 0, b, y, h->dense_mode);
    }
    for (int p = np - 1; p >= 0; --p) {
      const int j0 = p * DNP, nb = std::min(DNP, nc - j0), m = nc - j0 - nb, nch = (m + 63) / 64;
      if (nch > 0) hipLaunchKernelGGL(k_dn_bwdA, dim3(nch), dim3(DNS_THREADS), 0, st, nc, h->Sldl, j0, nb, h->xc, part, h->dense_mode);
      hipLaunchKernelGGL(k_dn_bwdB, dim3(1), dim3(DNS_THREADS), 0, st, nc, h->dn_z + (size_t)(2 * p + 1) * DNP * DNP, h->dvec, j0, nb, nch, y, part,
                         h->xc, h->dense_mode);
    }
  }
  if (nc > BK_THREADS && nc <= 1024)
    hipLaunchKernelGGL((k_coupling_solve<1024, 16>), dim3(1), dim3(1024), (size_t)nc * sizeof(double), st, nc, h->Sfac,
                       h->ipiv, h->Sldl, h->dvec, h->dense_mode, rc_dev, h->rs, h->xc, panels ? 1 : 0);
  else
    hipLaunchKernelGGL((k_coupling_solve<BK_THREADS, 32>), dim3(1), dim3(BK_THREADS), (size_t)nc * sizeof(double), st, nc,
                       h->Sfac, h->ipiv, h->Sldl, h->dvec, h->dense_mode, rc_dev, h->rs, h->xc);
  PP_HIP(hipGetLastError());
  return 0;
}

