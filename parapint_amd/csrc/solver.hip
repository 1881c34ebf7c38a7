// MI355X (gfx950) kernels and C-ABI of the batched Schur-complement KKT solver.
//
// Mapping (see plan.hpp): lane = scenario block.  Every value array is [entry][instance], so
// a wavefront touches 64 consecutive doubles (512 B) per access and all control flow / index
// data is wave-uniform (scalar loads, scalar branches): the sparse phase is a pure HBM/L2
// streaming workload with no divergence.  One 64-thread workgroup = one task x 64 instances.
//
// Kernels (reference function each one replaces):
//   k_transpose_in / k_assemble_sources   values of K_i, A_i -> panel storage (host / compact inputs; device-resident
//                                 sources are read by the leaf kernels themselves)   (MA27B input, ma27_interface.py:124)
//   k_gather_flat, k_gather_level_lean, k_scale_level   left-looking LDL^T in L form, static block pivots (MA27B)
//   (count workgroups of the Schur launch)   inertia / zero-pivot counts  (ma27_interface.py:201-203)
//   k_schur_mfma + k_schur_reduce_mfma (k_schur_tiles + k_scatter_schur for mapped groups)
//                                 S_local = -sum_i A_i K_i^-1 A_i^T     (mpi_explicit_schur_complement.py:312-333)
//   k_ldl_regs / k_ldl_blocked / k_dense_panel + k_dense_update, k_bk_factor   dense LDL^T of S + Q   (mpi_...:347-361)
//   k_bcr_ldl_inverse (k_bcr_factor, k_bcr_invert_wave), k_bcr_keep_y_mfma, k_bcr_update_mfma, k_corner_add
//                                 block-tridiagonal S of time-staged problems: cyclic reduction   (mpi_...:88-125, 352-361)
//   k_fwd_level(_pair), k_fwd_coupling, k_rs_reduce   forward substitution, r_s   (mpi_...:381-385; MA27C)
//   k_coupling_solve / k_bcr_fwd + k_bcr_bwd   x_c = S^-1 (r_c + r_s)   (mpi_...:388-391)
//   k_bwd_level(_pair)            back substitution with x_c            (mpi_...:393-396)
//   k_step_stats, k_vec_max_abs, k_vec_axpy   vector kernels of the interior-point step (interior_point.py:655-758)
#include <hip/hip_runtime.h>

#include <array>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <map>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/parapint_hip.h"
#include "dense_bk.hpp"
#include "plan.hpp"

namespace {

constexpr int WAVE = 64;
typedef double double4_t __attribute__((ext_vector_type(4)));   // accumulator of v_mfma_f64_16x16x4
constexpr int PP_MAX_SPLIT = 8;
constexpr int PP_MT_SLICE = 4;  // records (panel columns) of one 16 x 16 Schur tile per work item of k_schur_mfma
constexpr int PP_CSLOTS = 64;  // the inertia / growth counters are kept in this many slots of 4 ints, summed by the tail writer
constexpr int PP_TAIL = 8;    // doubles behind the n_c x n_c Schur block: zero pivots, pos, neg, host failures, growth, reserved
constexpr int PP_NPHASE = 8;  // assemble, factor, schur, dense, fwd, fwd_coupling, coupling_solve, bwd
constexpr int BK_THREADS = 512;
constexpr double PIVOT_EPS = 1e-13;
constexpr double BK_EPS = 1e-14;

// ------------------------------------------------------------------------------------------
// device image of one group's plan (all pointers are device memory)
struct GroupDev {
  int n, nc, batch, bpad, nchunk, npiv, nraw;
  int const_row;   // initial-value records that point at this row of the input are the constant 1 (f2 sources; -1: none)
  int64_t usize;
  const int *piv_w, *piv_start, *piv_uoff, *piv_doff, *piv_boff, *piv_sub, *piv_rowptr, *rowidx, *perm, *iperm;
  const int *piv_of_col, *rawmap, *raw_tiles;   // raw_tiles: 64-entry tiles of the input with at least one needed entry
  const int *ftask, *stask, *fdst_ptr, *fent;
  const int *clevel_col, *sfwd_eptr, *sfwd_upos, *sfwd_zcol;
  const int *fwd_rec, *bwd_rec;   // per scheduled column, in level order: everything its solve task needs (one scalar read)
  const int *crow_eptr, *crow_upos, *crow_zcol;
  const int *stile_a, *stile_b, *stile_ptr, *stile_rec;
  const int *mt_a, *mt_b, *mt_rec;   // 16 x 16 tiles of S for the MFMA form of the Schur update (unmapped groups): records of 32 row positions
  const int *mt_item, *mt_wptr;      // work items {first record, end} (slices of one tile's records), per tile its range of items
  double *raw, *rawT, *U, *L, *Dinv, *Tm, *Y, *X, *rhs, *xout, *Spart, *rspart;
  unsigned short* codes;
  const double* rhsN;   // right-hand sides in the native [row][instance] layout (caller order), or null: Y was filled by the transposition
  const int* cmapT; // mapped groups: global coupling index of local coupling row c of instance b at [c * bpad + b] (else null)
  double *Sloc, *XCL;   // mapped groups: per-instance Schur cliques [tile entry][instance], per-instance coupling solution
  int xs_row, xs_lane;  // address of coupling value c of lane b: c * xs_row + b * xs_lane (uniform: 1, 0 into xc)
  int* growth;      // per instance: 1 if a factor entry exceeded lbound (MA27's threshold test |l_ij| <= 1/u failed)
  double lbound;    // 1 / u_rt, or +inf
};

// ------------------------------------------------------------------------------------------
// Workgroup -> (task, chunk): one-dimensional grid with the 64-instance chunk as the fastest index.  Workgroups are
// dealt round-robin over the 8 XCDs, so with a chunk count that is a multiple of 8 every chunk is always served by
// the same XCD: the operands that different tasks of a level re-read for that chunk meet in ONE L2 instead of
// being duplicated in all eight.
#define PP_TASK_OF_WG(ny) ((int)(blockIdx.x / (unsigned)(ny)))
// Workgroups are dealt round-robin over the 8 XCDs, so workgroup x runs on XCD x mod 8.  Kernels with two instances
// per lane work on chunks of 128 instances (PP_PAIR_OF_WG: pair j = instance chunks 2j, 2j + 1, on XCD j mod 8); the
// kernels with one instance per lane must place the 64-instance chunks 2j and 2j + 1 on that same XCD, or every
// hand-over between the two kinds (gather -> scale -> gather, solve levels) crosses XCDs and misses its L2: within a
// run of 16 workgroups the chunk is 2 (s mod 8) + s / 8 instead of s.
__device__ __forceinline__ int pp_chunk64_perm(unsigned s, unsigned ny) {
  return (ny % 16u == 0u) ? (int)((s & ~15u) | ((s & 7u) << 1) | ((s >> 3) & 1u)) : (int)s;
}
__device__ __forceinline__ int pp_chunk64_of_wg(unsigned ny) { return pp_chunk64_perm(blockIdx.x % ny, ny); }
#define PP_CHUNK_OF_WG(ny) pp_chunk64_of_wg((unsigned)(ny))
#define PP_PAIR_OF_WG(ny) ((int)(blockIdx.x % (unsigned)(ny)))

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

// ------------------------------------------------------------------------------------------
// Record broadcast: the wave-uniform index records of a task are fetched with ONE coalesced
// vector load (lane e holds record e) and handed to all lanes with v_readlane, so the global
// loads of a whole task issue back to back (one memory latency) instead of being chained
// behind per-batch scalar loads.
__device__ __forceinline__ int bcast(int v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }
// 1/d without the IEEE division sequence: hardware estimate + two Newton steps (<= 1-2 ulp for the
// well-scaled pivots of the dense factor; not for denormal or near-overflow arguments)
__device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}
// wave-uniform broadcast of a double from a (wave-uniform) lane: two v_readlane_b32, no LDS crossbar
__device__ __forceinline__ double bcastd(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

// The same, but not before `dep` is available: ties the two v_readlane to a value of the consuming dependency
// chain.  The plain builtin is a pure function of a value loaded once, so all broadcasts of a long unrolled solve
// are hoisted to the top and their 2 x N SGPRs spilled to VGPR lanes and reloaded.
__device__ __forceinline__ double bcastd_after(double v, int src_lane, double dep) {
  int lo, hi;
  asm volatile("v_readlane_b32 %0, %2, %4\n\tv_readlane_b32 %1, %3, %4"
               : "=&s"(lo), "=&s"(hi)
               : "v"(__double2loint(v)), "v"(__double2hiint(v)), "s"(src_lane), "v"(dep));
  return __hiloint2double(hi, lo);
}

// inv(P) packed by rows of the lower triangle
#define PP_INV(inv, i, j) ((inv)[((i) > (j) ? (i) * ((i) + 1) / 2 + (j) : (j) * ((j) + 1) / 2 + (i))])

// Inversion of the gathered pivot block and scaling of panel rows [r0, r1): L rows = U rows * inv(P).
// Used by the scale tasks of big panels and as the closing phase of fused small-panel tasks.
// Task record (ints): piv, r0, r1, dptr0, kind, E0, E1, w, uoff, boff, doff, sub -- everything a task needs
// in one scalar read, so the entry records can be requested without first chasing the per-pivot arrays; gather
// tasks add piece, npieces (plan.hpp: a long row split over the waves of a quad).
constexpr int TASK_INTS = 16;

template <int WM>
__device__ __forceinline__ void invert_and_scale(const GroupDev& g, int p, int w, int uoff, int doff, unsigned sub,
                                                 int r0, int r1, double tmax_diag, bool publish, size_t bpad, int b,
                                                 double eps) {
  const double* Up = g.U + (size_t)uoff * bpad + b;
  double* Lp = g.L + (size_t)uoff * bpad + b;
  double inv[WM * (WM + 1) / 2];
  int code;
  if (WM == 1) {
    const pp::PivotResult pr = pp::invert_pivot(1, Up[0], 0.0, 0.0, tmax_diag, eps);
    inv[0] = pr.i00;
    code = (pr.code & 3) | (((pr.code >> 2) & 3) << 4) | (((pr.code >> 4) & 3) << 8);
  } else {
    double blk[WM * WM];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WM; ++j)
        blk[i * WM + j] = (i < w && j < w) ? Up[(size_t)(i * w + j) * bpad] : 0.0;
    code = pp::invert_block_t<WM>(w, sub, blk, tmax_diag, eps, inv);
  }
  if (publish) {
    double* invp = g.Dinv + (size_t)doff * bpad + b;
#pragma unroll
    for (int i = 0; i < WM * (WM + 1) / 2; ++i)
      if (i < w * (w + 1) / 2) invp[(size_t)i * bpad] = inv[i];
    g.codes[(size_t)p * bpad + b] = (b < g.batch) ? (unsigned short)code : (unsigned short)0;
  }
  bool grow = false;
  for (int r = (r0 > w ? r0 : w); r < r1; r += 4) {
    double u[4][WM];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t1 = 0; t1 < WM; ++t1)
        u[i][t1] = (r + i < r1 && t1 < w) ? Up[(size_t)((r + i) * w + t1) * bpad] : 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (r + i < r1) {
#pragma unroll
        for (int t2 = 0; t2 < WM; ++t2) {
          if (t2 < w) {
            double v = 0.0;
#pragma unroll
            for (int t1 = 0; t1 < WM; ++t1)
              if (t1 < w) v += u[i][t1] * PP_INV(inv, t1, t2);
            Lp[(size_t)((r + i) * w + t2) * bpad] = v;
            grow = grow || fabs(v) > g.lbound;
          }
        }
      }
    }
  }
  if (grow && b < g.batch) g.growth[b] = 1;
}

// One gather / fused task of the L-form factorisation (plan.hpp, FTask kinds 0 and 1): every
// destination row (all w columns of the block pivot) is acc[q] = -sum U[e.u] * L[e.l + q * e.wk]
// over its row entries, accumulated in registers and written once; the U operand is loaded once per
// entry and all global loads of a group of entries are in flight together.  Initial values come
// straight from the transposed input (e.u < 0, added to column e.q): assembly is fused into the
// factorisation.  Fused small panels (kind 1) finish with the inversion of their block and the
// scaling of their rows.
// NW = waves (tasks) per workgroup: PP_QUAD on the levels that hold split rows, 1 elsewhere (a workgroup keeps its
// resources until its longest wave ends, so unrelated tasks are better off as workgroups of their own).
// Written for few instructions per entry (round 2).  MEASURED at C3 (tools/pmc_metrics.sh): the round-1 kernel issued
// 1100 VALU + 1300 SALU instructions for ~20 entries and spent 56 % of its life waiting to issue (a 40 KB body of short
// branchy blocks), 23 % waiting for memory.  Here:
//   * row ends are marked in the records themselves (bits 8.. of the fourth field = rows that end before this entry),
//   * operands are addressed as uniform row base + lane offset, so the address arithmetic is scalar,
//   * the term magnitudes (zero-pivot test) are only tracked in the rows of the pivot block; all other rows are plain
//     fused multiply-adds,
//   * initial-value records load one operand, not 1 + w.
#ifdef PP_X_STAMPS
// diagnostic build: lane 0 of every wave of ONE launch (the level whose first task is pp_x_stamp_task0) writes 100 MHz
// timestamps at the stations of its task: [0] start, [1] task record read, [2] first entry records arrived, [3..]
// after each group of entries, [14] end, [15] hardware id
__device__ unsigned long long* pp_x_stamps = nullptr;
__device__ int pp_x_stamp_task0 = -1;
#define PP_STAMP(k) do { if (stp) stp[(k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PP_STAMP(k) do { } while (0)
#endif
// NV = instances per lane (1 or 2).  With 2, a lane owns the instances 2 * lane and 2 * lane + 1 of a 128-instance chunk:
// every operand request is one 16-byte load per lane, and the record broadcasts, the scalar address arithmetic and the
// branches of an entry are spent once for two instances (bpad must be a multiple of 128).
template <int NV>
__device__ __forceinline__ void ldv(const double* __restrict__ p, double (&out)[NV]) {
  if (NV == 1) out[0] = *p;
  else { const double2 t = *reinterpret_cast<const double2*>(p); out[0] = t.x; out[NV - 1] = t.y; }
}
template <int NV>
__device__ __forceinline__ void stv(double* p, const double (&v)[NV]) {
  if (NV == 1) *p = v[0];
  else *reinterpret_cast<double2*>(p) = make_double2(v[0], v[NV - 1]);
}

template <int WM, int NW, int NV>
__global__ __launch_bounds__(64 * NW) void k_gather_flat(GroupDev g, int task0, int chunk0, int ny, double eps) {
  __shared__ double red[NW > 1 ? NW : 1][NW > 1 ? 2 * WM * NV : 1][NW > 1 ? 64 : 1];   // partial sums / term magnitudes of a split row
  const int lane = threadIdx.x & 63, wave = (NW > 1) ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
  const unsigned b = (unsigned)((((NV == 2 ? PP_PAIR_OF_WG(ny) : PP_CHUNK_OF_WG(ny)) + chunk0) * 64 + lane) * NV);    // first instance of this lane
  const size_t bpad = (size_t)g.bpad;
  const int* t = g.ftask + TASK_INTS * (size_t)(task0 + NW * PP_TASK_OF_WG(ny) + wave);
#ifdef PP_X_STAMPS
  unsigned long long* stp = (pp_x_stamps && task0 == pp_x_stamp_task0 && lane == 0)
                                ? pp_x_stamps + 16 * ((size_t)blockIdx.x * NW + wave) : nullptr;
  int stamp_k = 3;
  PP_STAMP(0);
#endif
  const int p = t[0], r0 = t[1], r1 = t[2], kind = t[4], E0 = t[5], E1 = t[6];
  const int piece = (NW > 1) ? t[12] : 0, npieces = (NW > 1) ? t[13] : 1;   // npieces is the same for all waves of the workgroup
  if (kind < 0 && npieces <= 1) return;            // quad padding (in a split quad the padding waves join the barrier)
  const int w = (WM == 1) ? 1 : t[7];
  const int uoff = t[8], boff = t[9], doff = t[10];
  const unsigned sub = (unsigned)t[11];
  const int wp = t[14], qoff = t[15];     // width of the whole panel and first column of this task's slice (root front; else w, 0)
#ifdef PP_X_STAMPS
  if (stp) { stp[1] = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(E1 < 0); stp[13] = (unsigned long long)(E1 - E0); }
#endif
  const double* __restrict__ Ub = g.U;
  const double* __restrict__ Lb = g.L;
  const double* __restrict__ Rb = g.rawT;
  const int nrow = r1 - r0;
  double* Udst = g.U + ((size_t)uoff + (size_t)r0 * wp + qoff) * bpad;     // uniform; lane offset added at the store
  double* Tmd = g.Tm + ((size_t)boff + (size_t)r0 * wp + qoff) * bpad;
  const int nblk = (r0 < wp) ? (wp - r0) : 0;        // leading destination rows that belong to the pivot block
  const bool split = NW > 1 && npieces > 1;
  double tmax_diag[NV];
  double acc[WM][NV], tmax[WM][NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    tmax_diag[v] = 0.0;
#pragma unroll
    for (int q = 0; q < WM; ++q) { acc[q][v] = 0.0; tmax[q][v] = 0.0; }
  }
  int d = 0;
  auto finalize = [&]() {
    if (!split) {                                    // (split row: combined below)
#pragma unroll
      for (int q = 0; q < WM; ++q)
        if (q < w) stv<NV>(Udst + (size_t)(d * wp + q) * bpad + b, acc[q]);
      if (d < nblk) {
#pragma unroll
        for (int q = 0; q < WM; ++q) {
          if (q < w) {
            if (kind == 0) stv<NV>(Tmd + (size_t)(d * wp + q) * bpad + b, tmax[q]);
            else {
#pragma unroll
              for (int v = 0; v < NV; ++v) tmax_diag[v] = fmax(tmax_diag[v], tmax[q][v]);
            }
          }
#pragma unroll
          for (int v = 0; v < NV; ++v) tmax[q][v] = 0.0;
        }
      }
#pragma unroll
      for (int q = 0; q < WM; ++q)
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[q][v] = 0.0;
    }
    ++d;
  };
  constexpr int G = (WM == 1) ? 8 : 4;               // entries whose operands are requested together
  for (int eb = E0; eb < E1; eb += 64) {
    const int cnt = min(64, E1 - eb);
    int4 rec = make_int4(0, 0, 0, 0);
    if (lane < cnt) rec = *reinterpret_cast<const int4*>(g.fent + 4 * (size_t)(eb + lane));
#ifdef PP_X_STAMPS
    if (stp && eb == E0) stp[2] = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(bcast(rec.x, 0) == 0x7fffffff);
#endif
    for (int i0 = 0; i0 < cnt; i0 += G) {
      int eu[G], el[G], ew[G], ef[G];
      double su[G][NV], sl[G][WM][NV];
#pragma unroll
      for (int i = 0; i < G; ++i) {
        const int qi = min(i0 + i, cnt - 1);
        eu[i] = bcast(rec.x, qi); el[i] = bcast(rec.y, qi); ew[i] = bcast(rec.z, qi); ef[i] = bcast(rec.w, qi);
      }
      // branch-free operand requests (a branch here splits the requests over basic blocks, and the wait-count insertion
      // then drains all outstanding loads at the joins: two or more round trips per group instead of one).  An
      // initial-value record requests row 0 of L for its unused operands.
#pragma unroll
      for (int i = 0; i < G; ++i) {
        const bool prod = eu[i] >= 0;
        const int idx = prod ? eu[i] : -1 - eu[i];
        const double* __restrict__ base = prod ? Ub : Rb;
        ldv<NV>(base + (size_t)((!prod && idx == g.const_row) ? 0 : idx) * bpad + b, su[i]);
#pragma unroll
        for (int q = 0; q < WM; ++q)
          ldv<NV>(Lb + (size_t)(prod ? el[i] + min(q, w - 1) * ew[i] : 0) * bpad + b, sl[i][q]);
      }
#pragma unroll
      for (int i = 0; i < G; ++i) {
        if (i0 + i < cnt) {
          for (int nf = ef[i] >> 8; nf > 0; --nf) finalize();
          if (eu[i] >= 0) {
            if (ew[i] == 0) {
              // single-column entry (the source panel holds only some columns of this block pivot as rows): all w
              // requests above went to the one L operand
              const int eq = ef[i] & 0xff;
#pragma unroll
              for (int q = 0; q < WM; ++q) {
                if (q == eq) {
#pragma unroll
                  for (int v = 0; v < NV; ++v) {
                    const double term = su[i][v] * sl[i][q][v];
                    acc[q][v] -= term;
                    if (d < nblk) tmax[q][v] = fmax(tmax[q][v], fabs(term));
                  }
                }
              }
            } else if (d < nblk) {
#pragma unroll
              for (int q = 0; q < WM; ++q)
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                  const double term = su[i][v] * ((q < w) ? sl[i][q][v] : 0.0);
                  acc[q][v] -= term;
                  tmax[q][v] = fmax(tmax[q][v], fabs(term));
                }
            } else {
#pragma unroll
              for (int q = 0; q < WM; ++q)     // (q >= w: a duplicate of column w - 1, never stored)
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[q][v] = fma(-su[i][v], sl[i][q][v], acc[q][v]);
            }
          } else {
            // initial-value record: (y, z) hold the coefficient of the input entry (1 for plain raw values)
            const bool cst = (-1 - eu[i] == g.const_row);
            const double coef = __hiloint2double(ew[i], el[i]);
            const int eq = ef[i] & 0xff;
#pragma unroll
            for (int q = 0; q < WM; ++q) {
              if (q == eq) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                  const double term = (cst ? 1.0 : su[i][v]) * coef;
                  acc[q][v] += term;
                  tmax[q][v] = fmax(tmax[q][v], fabs(term));
                }
              }
            }
          }
        }
      }
#ifdef PP_X_STAMPS
      if (stp && stamp_k < 13) { stp[stamp_k] = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(acc[0][0] == 1.2345e300); ++stamp_k; }
#endif
    }
  }
  if (split) {
    // one long row over the waves of this quad: partial sums meet in LDS, piece 0 adds them in piece order
#pragma unroll
    for (int q = 0; q < WM; ++q)
#pragma unroll
      for (int v = 0; v < NV; ++v) { red[wave][q * NV + v][lane] = acc[q][v]; red[wave][(WM + q) * NV + v][lane] = tmax[q][v]; }
    __syncthreads();
    if (piece == 0 && kind >= 0) {
#pragma unroll
      for (int q = 0; q < WM; ++q) {
        double a[NV], m[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          a[v] = acc[q][v]; m[v] = tmax[q][v];
          for (int j = 1; j < npieces; ++j) { a[v] += red[j][q * NV + v][lane]; m[v] = fmax(m[v], red[j][(WM + q) * NV + v][lane]); }
        }
        if (q < w) {
          stv<NV>(Udst + (size_t)q * bpad + b, a);
          if (r0 < wp) stv<NV>(Tmd + (size_t)q * bpad + b, m);
        }
      }
    }
    return;
  }
  while (d < nrow) finalize();
  if (kind == 1) {
#pragma unroll
    for (int v = 0; v < NV; ++v)
      invert_and_scale<WM>(g, p, w, uoff, doff, sub, r0, r1, tmax_diag[v], true, bpad, (int)b + v, eps);
  }
#ifdef PP_X_STAMPS
  if (stp) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stp[14] = __builtin_amdgcn_s_memrealtime();
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    stp[15] = hw;
  }
#endif
}

// Lean variant for the wide bottom levels of the tree (a few entries per task, tens of thousands
// of tasks): plain scalar-load record reads, minimal code; the memory system is kept busy by the
// sheer number of waves, not by intra-task batching.
template <int WM, int NV>
__global__ __launch_bounds__(64) void k_gather_level_lean(GroupDev g, int task0, int chunk0, int ny, double eps) {
  const int lane = threadIdx.x;
  const unsigned b = (unsigned)((((NV == 2 ? PP_PAIR_OF_WG(ny) : PP_CHUNK_OF_WG(ny)) + chunk0) * 64 + lane) * NV);    // first instance of this lane
  const size_t bpad = (size_t)g.bpad;
  const int* t = g.ftask + TASK_INTS * (size_t)(task0 + PP_TASK_OF_WG(ny));
  const int p = t[0], r0 = t[1], r1 = t[2], dptr0 = t[3], kind = t[4];
  if (kind < 0) return;                            // quad padding (lean levels have no split rows)
  const int w = (WM == 1) ? 1 : t[7];
  const int uoff = t[8], boff = t[9], doff = t[10];
  const unsigned sub = (unsigned)t[11];
  const int wp = t[14], qoff = t[15];     // (root front: panel width, first column of the slice; else w, 0)
  const double* __restrict__ U = g.U + b;
  const double* __restrict__ Lb = g.L + b;
  const double* __restrict__ R = g.rawT + b;
  const int nrow = r1 - r0;
  const int* dp = g.fdst_ptr + dptr0;
  double* Udst = g.U + ((size_t)uoff + (size_t)r0 * wp + qoff) * bpad + b;
  double* Tmd = g.Tm + ((size_t)boff + (size_t)r0 * wp + qoff) * bpad + b;
  double* Ldst = g.L + ((size_t)uoff + (size_t)r0 * wp + qoff) * bpad + b;
  const int nblk = (r0 < wp) ? (wp - r0) : 0;
  double tmax_diag[NV], inv1[NV], lmax = 0.0;
#pragma unroll
  for (int v = 0; v < NV; ++v) { tmax_diag[v] = 0.0; inv1[v] = 0.0; }
  for (int d = 0; d < nrow; ++d) {
    double acc[WM][NV], tmax[WM][NV];
#pragma unroll
    for (int q = 0; q < WM; ++q)
#pragma unroll
      for (int v = 0; v < NV; ++v) { acc[q][v] = 0.0; tmax[q][v] = 0.0; }
    for (int e = dp[d]; e < dp[d + 1]; ++e) {
      // scalar (SMEM) record reads; the operand base is chosen by offset, not by pointer select
      const int* rp = g.fent + 4 * (size_t)e;
      const int ex = rp[0], ey = rp[1], ez = rp[2], ew = rp[3];
      const bool cst = ex < 0 && (-1 - ex) == g.const_row;
      double sv[NV];
      ldv<NV>((ex >= 0) ? U + (size_t)ex * bpad : R + (size_t)(cst ? 0 : -1 - ex) * bpad, sv);
      const double coef = __hiloint2double(ez, ey);     // (initial-value records: coefficient of the input entry)
#pragma unroll
      for (int q = 0; q < WM; ++q) {
        double lv[NV];
        ldv<NV>(Lb + (size_t)((ex >= 0) ? ey + min(q, w - 1) * ez : 0) * bpad, lv);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          // (product entry over all w columns; single-column product entry, ez == 0; initial value)
          const double m = (ex >= 0) ? ((ez != 0 ? q < w : q == (ew & 0xff)) ? lv[v] : 0.0) : ((q == (ew & 0xff)) ? -coef : 0.0);
          const double term = (cst ? 1.0 : sv[v]) * m;
          acc[q][v] -= term;
          tmax[q][v] = fmax(tmax[q][v], fabs(term));
        }
      }
    }
#pragma unroll
    for (int q = 0; q < WM; ++q)
      if (q < w) stv<NV>(Udst + (size_t)(d * wp + q) * bpad, acc[q]);
    if (WM == 1 && kind == 1) {
      // scalar pivot, fused panel: row 0 is the pivot, every later one a row to scale
      if (d == 0) {
        unsigned short code[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const pp::PivotResult pr = pp::invert_pivot(1, acc[0][v], 0.0, 0.0, tmax[0][v], eps);
          inv1[v] = pr.i00;
          const int c = (pr.code & 3) | (((pr.code >> 2) & 3) << 4) | (((pr.code >> 4) & 3) << 8);
          code[v] = ((int)(b + v) < g.batch) ? (unsigned short)c : (unsigned short)0;
        }
        stv<NV>(g.Dinv + (size_t)doff * bpad + b, inv1);
        if (NV == 1) g.codes[(size_t)p * bpad + b] = code[0];
        else *reinterpret_cast<unsigned int*>(g.codes + (size_t)p * bpad + b) = (unsigned int)code[0] | ((unsigned int)code[NV - 1] << 16);
      } else {
        double lv[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          lv[v] = acc[0][v] * inv1[v];
          lmax = fmax(lmax, ((int)(b + v) < g.batch) ? fabs(lv[v]) : 0.0);   // (one flag store per task, below: a store per row cost 36 us per step at C3)
        }
        stv<NV>(Ldst + (size_t)d * bpad, lv);
      }
    } else if (d < nblk) {
#pragma unroll
      for (int q = 0; q < WM; ++q) {
        if (q < w) {
          if (kind == 0) stv<NV>(Tmd + (size_t)(d * wp + q) * bpad, tmax[q]);
          else {
#pragma unroll
            for (int v = 0; v < NV; ++v) tmax_diag[v] = fmax(tmax_diag[v], tmax[q][v]);
          }
        }
      }
    }
  }
  if (WM == 1 && lmax > g.lbound) {
    // (which of the two instances of the lane grew is not kept: both are flagged; the guard re-orders from either)
#pragma unroll
    for (int v = 0; v < NV; ++v) if ((int)(b + v) < g.batch) g.growth[b + v] = 1;
  }
  if (WM != 1 && kind == 1) {
#pragma unroll
    for (int v = 0; v < NV; ++v)
      invert_and_scale<WM>(g, p, w, uoff, doff, sub, r0, r1, tmax_diag[v], true, bpad, (int)b + v, eps);
  }
}

// L values [v0, v1) of a panel of compile-time width W from row values held in registers
template <int W, int RV>
__device__ __forceinline__ bool scale_held(const double (&u)[RV], const double* inv, double* Lp, int v0, int v1,
                                           size_t bpad, double lbound) {
  bool grow = false;
#pragma unroll
  for (int i = 0; i < RV; ++i) {
    if (v0 + i < v1) {
      constexpr int dummy = 0; (void)dummy;
      const int t2 = i % W, ib = i - i % W;
      double v = 0.0;
#pragma unroll
      for (int t1 = 0; t1 < W; ++t1) v += u[(ib + t1) < RV ? (ib + t1) : RV - 1] * PP_INV(inv, t1, t2);
      Lp[(size_t)(v0 + i) * bpad] = v;
      grow = grow || fabs(v) > lbound;
    }
  }
  return grow;
}

// Scale task of a big panel (plan.hpp, kind 2): invert the gathered pivot block (every chunk does it
// redundantly in registers -- it is w*w loads and a few dozen flops), L rows = U rows * inv(P); the
// chunk that starts right below the block also publishes inv(P) and the inertia code.
template <int WM>
__global__ __launch_bounds__(64) void k_scale_level(GroupDev g, int task0, int chunk0, int ny, double eps) {
  const int lane = threadIdx.x;
  const int b = (PP_CHUNK_OF_WG(ny) + chunk0) * 64 + lane;
  const size_t bpad = (size_t)g.bpad;
  const int* t = g.stask + TASK_INTS * (size_t)(task0 + PP_TASK_OF_WG(ny));
  const int p = t[0], r0 = t[1], r1 = t[2], w = t[7], uoff = t[8], boff = t[9], doff = t[10];
  const unsigned sub = (unsigned)t[11];
  const double* Tmp = g.Tm + (size_t)boff * bpad + b;
  const double* Up = g.U + (size_t)uoff * bpad + b;
  double* Lp = g.L + (size_t)uoff * bpad + b;
  // block, its term magnitudes and the first rows of the chunk are requested together (one round trip)
  constexpr int RV = 8;   // row values (rows * w) held while the block is inverted
  double tm[WM * WM], blk[WM * WM], u[RV];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WM; ++j) {
      const bool in = i < w && j < w;
      const size_t off = (size_t)(in ? i * w + j : 0) * bpad;
      const double tv = Tmp[off], uv = Up[off];
      tm[i * WM + j] = in ? tv : 0.0;
      blk[i * WM + j] = in ? uv : 0.0;
    }
  const int v0 = r0 * w, v1 = r1 * w;      // value range [v0, v1) of this chunk in the panel
#pragma unroll
  for (int i = 0; i < RV; ++i) u[i] = Up[(size_t)min(v0 + i, v1 - 1) * bpad];
  double tmax_diag = 0.0;
#pragma unroll
  for (int i = 0; i < WM * WM; ++i) tmax_diag = fmax(tmax_diag, tm[i]);
  double inv[WM * (WM + 1) / 2];
  const int code = pp::invert_block_t<WM>(w, sub, blk, tmax_diag, eps, inv);
  if (r0 == w) {
    double* invp = g.Dinv + (size_t)doff * bpad + b;
#pragma unroll
    for (int i = 0; i < WM * (WM + 1) / 2; ++i)
      if (i < w * (w + 1) / 2) invp[(size_t)i * bpad] = inv[i];
    g.codes[(size_t)p * bpad + b] = (b < g.batch) ? (unsigned short)code : (unsigned short)0;
  }
  if (v1 - v0 <= RV) {
    // the common shapes (8 rows x 1, 4 x 2, 2 x 4): rows already in registers
    bool gr = false, done = true;
    if (w == 1) gr = scale_held<1, RV>(u, inv, Lp, v0, v1, bpad, g.lbound);
    else if (w == 2) gr = scale_held<2, RV>(u, inv, Lp, v0, v1, bpad, g.lbound);
    else if (w == 4) gr = scale_held<4, RV>(u, inv, Lp, v0, v1, bpad, g.lbound);
    else if (WM >= 8 && w == 8) gr = scale_held<(WM >= 8 ? 8 : 1), RV>(u, inv, Lp, v0, v1, bpad, g.lbound);
    else done = false;
    if (done) {
      if (gr && b < g.batch) g.growth[b] = 1;
      return;
    }
  }
  bool grow = false;
  for (int r = r0; r < r1; ++r) {
    double ur[WM];
#pragma unroll
    for (int t1 = 0; t1 < WM; ++t1) ur[t1] = (t1 < w) ? Up[(size_t)(r * w + t1) * bpad] : 0.0;
#pragma unroll
    for (int t2 = 0; t2 < WM; ++t2) {
      if (t2 < w) {
        double v = 0.0;
#pragma unroll
        for (int t1 = 0; t1 < WM; ++t1)
          if (t1 < w) v += ur[t1] * PP_INV(inv, t1, t2);
        Lp[(size_t)(r * w + t2) * bpad] = v;
        grow = grow || fabs(v) > g.lbound;
      }
    }
  }
  if (grow && b < g.batch) g.growth[b] = 1;
}

// ------------------------------------------------------------------------------------------
// Root front (plan.hpp, front_piv): the last block pivot, up to PP_FRONT_MAX columns wide.  Its rows were gathered in
// column slices by the ordinary tasks; k_front_invert inverts the w x w pivot block with the static sequence of
// 1x1 / 2x2 sub-pivots (pivot.hpp: invert_front is the definition) and k_scale_wide forms L = U inv(P).
// One workgroup per chunk of 64 instances, lane = instance, wave i = row i of A (16 waves): per sub-pivot the owner(s)
// of the pivot row(s) test and invert the pivot and publish the OLD row(s) and the inverse through LDS; every other
// wave updates its row from them, the owners scale theirs.
struct FrontRec { int piv, w, uoff, boff, doff; unsigned sub; };
__global__ __launch_bounds__(512) void k_front_invert(GroupDev g, FrontRec fr, double* __restrict__ finv, double eps) {
  constexpr int WF = pp::PP_WF, RW = 2, NWV = WF / RW;   // rows of A per wave, waves
  __shared__ double rowk[2][2][WF][64];   // [step parity][first / second pivot row][column][lane]: the pivot rows before the step
  __shared__ double pinv[2][3][64];       // [step parity]: i00, i10, i11
  __shared__ double red[NWV][64];
  __shared__ int cnt[NWV][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * 64 + lane;
  const size_t bpad = (size_t)g.bpad;
  const int w = fr.w;
  double A[RW][WF];                       // rows RW wave .. of the block (all columns, both triangles)
  int codes = 0;                          // inertia counts of the pivots this wave owned: pos | neg << 8 | zero << 16
  double tm = 0.0;
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int j = 0; j < WF; ++j) {
      const int i = RW * wave + r;
      const bool in = i < w && j < w;
      const int hi = i > j ? i : j, lo = i > j ? j : i;       // (the lower triangle of the gathered block is the matrix)
      const size_t off = (size_t)(in ? hi * w + lo : 0) * bpad + b;
      const double av = g.U[(size_t)fr.uoff * bpad + off], tv = g.Tm[(size_t)fr.boff * bpad + off];
      A[r][j] = in ? av : 0.0;
      tm = fmax(tm, in ? tv : 0.0);
    }
#ifdef PP_X_STAMPS
  unsigned long long* stp = (pp_x_stamps && lane == 0) ? pp_x_stamps + 4000000 + 16 * ((size_t)blockIdx.x * NWV + wave) : nullptr;
  int stamp_k = 2;
  if (stp) stp[0] = __builtin_amdgcn_s_memrealtime();
#endif
  red[wave][lane] = tm;
  __syncthreads();
  double colmax = 0.0;
#pragma unroll
  for (int i = 0; i < NWV; ++i) colmax = fmax(colmax, red[i][lane]);
#ifdef PP_X_STAMPS
  if (stp) stp[1] = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(colmax == 1.2345e300);
#endif
  // The step loop is unrolled: every register index below is a constant (a rolled loop selects the pivot column with
  // compare / select pairs per element: 1.75 us per step, measured).  One barrier per 1x1 step: the buffers of a
  // step are written again two steps later, which every wave reaches only through the barrier of the step in between.
  bool second = false;
#pragma unroll
  for (int k = 0; k < PP_FRONT_MAX; ++k) {
    if (k >= w) continue;                 // (uniform)
    if (second) { second = false; continue; }
    const int par = k & 1;
    constexpr int dummy = 0; (void)dummy;
    const int k1 = (k + 1 < WF) ? k + 1 : k;
    const int ow = k / RW, orow = k % RW, ow1 = k1 / RW, orow1 = k1 % RW;     // owners of the pivot rows (constants)
    const bool two = ((fr.sub >> k) & 1u) && (k + 1 < w);
    if (!two) {
      if (wave == ow) {
#pragma unroll
        for (int j = 0; j < WF; ++j) rowk[par][0][j][lane] = A[orow][j];
        const pp::PivotResult pr = pp::invert_pivot(1, A[orow][k], 0.0, 0.0, colmax, eps);
        pinv[par][0][lane] = pr.i00;
        codes += (pr.code & 3) | (((pr.code >> 2) & 3) << 8) | (((pr.code >> 4) & 3) << 16);
      }
      __syncthreads();
      const double i00 = pinv[par][0][lane];
      double rk[WF];
#pragma unroll
      for (int j = 0; j < WF; ++j) rk[j] = rowk[par][0][j][lane];
#pragma unroll
      for (int r = 0; r < RW; ++r) {      // (the owner's pivot row is overwritten below)
        const double l = A[r][k] * i00;
#pragma unroll
        for (int j = 0; j < WF; ++j)
          if (j != k) A[r][j] -= l * rk[j];
        A[r][k] = l;
      }
      if (wave == ow) {
#pragma unroll
        for (int j = 0; j < WF; ++j) A[orow][j] = (j == k) ? -i00 : rk[j] * i00;
      }
    } else {
      second = true;
      if (wave == ow) {
#pragma unroll
        for (int j = 0; j < WF; ++j) rowk[par][0][j][lane] = A[orow][j];
      }
      if (wave == ow1) {
#pragma unroll
        for (int j = 0; j < WF; ++j) rowk[par][1][j][lane] = A[orow1][j];
      }
      __syncthreads();
      if (wave == ow) {       // (a, b, c) = A[k][k], A[k1][k], A[k1][k1]
        const pp::PivotResult pr = pp::invert_pivot(2, A[orow][k], rowk[par][1][k][lane], rowk[par][1][k1][lane], colmax, eps);
        pinv[par][0][lane] = pr.i00; pinv[par][1][lane] = pr.i10; pinv[par][2][lane] = pr.i11;
        codes += (pr.code & 3) | (((pr.code >> 2) & 3) << 8) | (((pr.code >> 4) & 3) << 16);
      }
      __syncthreads();
      const double i00 = pinv[par][0][lane], i10 = pinv[par][1][lane], i11 = pinv[par][2][lane];
      double rk[WF], rk1[WF];
#pragma unroll
      for (int j = 0; j < WF; ++j) { rk[j] = rowk[par][0][j][lane]; rk1[j] = rowk[par][1][j][lane]; }
#pragma unroll
      for (int r = 0; r < RW; ++r) {      // (the owners' pivot rows are overwritten below)
        const double l0 = A[r][k] * i00 + A[r][k1] * i10;
        const double l1 = A[r][k] * i10 + A[r][k1] * i11;
#pragma unroll
        for (int j = 0; j < WF; ++j)
          if (j != k && j != k1) A[r][j] -= l0 * rk[j] + l1 * rk1[j];
        A[r][k] = l0; A[r][k1] = l1;
      }
      if (wave == ow) {
#pragma unroll
        for (int j = 0; j < WF; ++j) A[orow][j] = (j == k) ? -i00 : (j == k1) ? -i10 : rk[j] * i00 + rk1[j] * i10;
      }
      if (wave == ow1) {
#pragma unroll
        for (int j = 0; j < WF; ++j) A[orow1][j] = (j == k) ? -i10 : (j == k1) ? -i11 : rk[j] * i10 + rk1[j] * i11;
      }
    }
#ifdef PP_X_STAMPS
    if (stp && stamp_k < 15) { stp[stamp_k++] = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(A[0][0] == 1.2345e300); }
#endif
  }
  {
    // inv(P) = -A: packed by rows of the lower triangle for the solve sweeps, and as a full 16 x 16 matrix (zero beyond
    // w) for k_scale_wide, whose operand addresses are then constants
    double* invp = g.Dinv + (size_t)fr.doff * bpad + b;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int i = RW * wave + r;
#pragma unroll
      for (int j = 0; j < WF; ++j) {
        if (i < w && j <= i) invp[(size_t)(i * (i + 1) / 2 + j) * bpad] = -A[r][j];
        finv[(size_t)(i * WF + j) * bpad + b] = (i < w && j < w) ? -A[r][j] : 0.0;
      }
    }
  }
  cnt[wave][lane] = codes;
  __syncthreads();
  if (wave == 0) {
    int c = 0;
#pragma unroll
    for (int i = 0; i < NWV; ++i) c += cnt[i][lane];
    const int code = (c & 255) | (((c >> 8) & 255) << 4) | (((c >> 16) & 255) << 8);
    g.codes[(size_t)fr.piv * bpad + b] = (b < g.batch) ? (unsigned short)code : (unsigned short)0;
  }
#ifdef PP_X_STAMPS
  if (stp) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stp[15] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

// Rows [r0, r1) of the root front: L rows = U rows * inv(P) with the explicit inverse from k_front_invert (the full,
// zero-padded 16 x 16 form).  One workgroup of three waves per (chunk of rows, chunk of instances); wave c holds columns
// 5c .. 5c + 4 of inv(P) in registers (75 values) and forms those entries of every row of the chunk.
__global__ __launch_bounds__(192) void k_scale_wide(GroupDev g, const int* __restrict__ wtask, FrontRec fr,
                                                    const double* __restrict__ finv, int ny) {
  constexpr int WF = PP_FRONT_MAX, CW = 5;
  const int lane = threadIdx.x & 63, cg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = PP_CHUNK_OF_WG(ny) * 64 + lane;
  const size_t bpad = (size_t)g.bpad;
  const int* t = wtask + TASK_INTS * (size_t)PP_TASK_OF_WG(ny);
  const int r0 = t[1], r1 = t[2], w = fr.w;
  const int c0 = CW * cg;
  if (c0 >= w) return;
#ifdef PP_X_STAMPS
  unsigned long long* stp = (pp_x_stamps && lane == 0) ? pp_x_stamps + 4500000 + 16 * ((size_t)blockIdx.x * 3 + cg) : nullptr;
  if (stp) stp[0] = __builtin_amdgcn_s_memrealtime();
#endif
  double ic[WF][CW];
  {
    const double* p = finv + (size_t)c0 * bpad + b;      // row t1 of the 16 x 16 matrix, columns c0 ..
#pragma unroll
    for (int t1 = 0; t1 < WF; ++t1) {
#pragma unroll
      for (int q = 0; q < CW; ++q) ic[t1][q] = p[(size_t)q * bpad];
      p += (size_t)pp::PP_WF * bpad;
    }
  }
  const double* Up = g.U + (size_t)fr.uoff * bpad + b;
  double* Lp = g.L + (size_t)fr.uoff * bpad + b;
  double lmax = 0.0;
#ifdef PP_X_STAMPS
  if (stp) stp[1] = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(ic[0][0] == 1.2345e300);
  int stamp_k = 2;
#endif
  for (int r = r0; r < r1; ++r) {
#ifdef PP_X_STAMPS
    if (stp && stamp_k < 15) stp[stamp_k++] = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(lmax == 1.2345e300);
#endif
    double u[WF];
#pragma unroll
    for (int t1 = 0; t1 < WF; ++t1) u[t1] = Up[(size_t)(r * w + min(t1, w - 1)) * bpad];    // (columns >= w meet zero rows of inv)
    double v[CW];
#pragma unroll
    for (int q = 0; q < CW; ++q) v[q] = 0.0;
#pragma unroll
    for (int t1 = 0; t1 < WF; ++t1)
#pragma unroll
      for (int q = 0; q < CW; ++q) v[q] += u[t1] * ic[t1][q];
#pragma unroll
    for (int q = 0; q < CW; ++q)
      if (c0 + q < w) { Lp[(size_t)(r * w + c0 + q) * bpad] = v[q]; lmax = fmax(lmax, fabs(v[q])); }
  }
  if (lmax > g.lbound && b < g.batch) g.growth[b] = 1;
#ifdef PP_X_STAMPS
  if (stp) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stp[15] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

// counters[0..2] += (pos, neg, zero) over all block pivots (codes of padded instances are 0);
// a code is pos | neg << 4 | zero << 8 in 16 bits, 8 codes per 16-byte load
__global__ __launch_bounds__(256) void k_count_codes(const unsigned short* __restrict__ codes, size_t total8,
                                                     int* counters, int* __restrict__ growth, int* __restrict__ growth_seen,
                                                     int batch) {
  __shared__ int red[3][256];
  int pos = 0, neg = 0, zero = 0;
  if (blockIdx.x == 0) {     // instances whose factor showed element growth beyond 1 / u_rt: counted, kept for
    int gr = 0;              // pp_find_growth and cleared for the next factorisation
    for (int i = threadIdx.x; i < batch; i += 256) {
      const int f = growth[i];
      gr += f != 0;
      growth_seen[i] = f;
      if (f) growth[i] = 0;
    }
    if (gr) atomicAdd(&counters[3], gr);
  }
  const uint4* c4 = reinterpret_cast<const uint4*>(codes);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (size_t)gridDim.x * 256) {
    const uint4 v = c4[i];
    const unsigned int wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned int x = wds[q];
      pos += (int)((x & 15u) + ((x >> 16) & 15u));
      neg += (int)(((x >> 4) & 15u) + ((x >> 20) & 15u));
      zero += (int)(((x >> 8) & 15u) + ((x >> 24) & 15u));
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

// Status mailbox: block counters (S tail after the all-reduce) + dense-factor counters -> pinned host
// memory, sequence word last (LinearSolverStatus / get_inertia read-back, mpi_...:19-30, 417-436).
__device__ __forceinline__ void publish_status(const double* __restrict__ tail, const int* __restrict__ bk,
                                               long long* out, long long seq) {
  const long long zero = (long long)(tail[0] + 0.5) + bk[2];
  out[1] = (long long)(tail[1] + 0.5) + bk[0];
  out[2] = (long long)(tail[2] + 0.5) + bk[1];
  out[3] = zero;
  // tail[3]: host-side failures of any rank, summed by the all-reduce (pp_fail_local: 1 per not_enough_memory,
  // 1e3 per singular, 1e6 per error); the most severe status wins (error > singular > not_enough_memory)
  const double hs = tail[3];
  const long long growth = (long long)(tail[4] + 0.5);
  out[5] = growth;
  long long st = zero > 0 ? 2 : 0;     // (growth: pp_get_status decides, it knows whether the guard is enforced)
  if (hs >= 1e6) st = 3;
  else if (hs >= 1e3) st = 2;
  else if (hs >= 1.0 && st == 0) st = 1;
  out[0] = st;
  __threadfence_system();
  __hip_atomic_store(out + 4, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// (n_c = 0: no dense phase; otherwise the last dense kernel, k_bk_factor, publishes)
__global__ void k_publish_status(const double* __restrict__ tail, const int* __restrict__ bk, long long* out,
                                 long long seq) {
  if (threadIdx.x == 0 && blockIdx.x == 0) publish_status(tail, bk, out, seq);
}

// ------------------------------------------------------------------------------------------
// Inertia codes and growth flags of the group, counted by `ncb` 64-thread workgroups (this is workgroup `cb` of them) into
// the slotted counters; runs in front of the Schur tile workgroups of the same launch.
__device__ __forceinline__ void count_codes_block(const GroupDev& g, unsigned cb, unsigned ncb, size_t total8, int* counters, int lane) {
  int cnt[4] = {0, 0, 0, 0};     // pos, neg, zero, growth
  if (cb == 0) {
    int* growth_seen = g.growth + g.bpad;
    for (int i = lane; i < g.batch; i += 64) {
      const int f = g.growth[i];
      cnt[3] += f != 0;
      growth_seen[i] = f;
      if (f) g.growth[i] = 0;
    }
  }
  const uint4* c4 = reinterpret_cast<const uint4*>(g.codes);
  const size_t stride = (size_t)ncb * 64;
  for (size_t i = (size_t)cb * 64 + lane; i < total8; i += 4 * stride) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = (i + u * stride < total8) ? c4[i + u * stride] : make_uint4(0, 0, 0, 0);   // in flight together
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned int wds[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned int x = wds[q];
        cnt[0] += (int)((x & 15u) + ((x >> 16) & 15u));
        cnt[1] += (int)(((x >> 4) & 15u) + ((x >> 20) & 15u));
        cnt[2] += (int)(((x >> 8) & 15u) + ((x >> 24) & 15u));
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    for (int off = 32; off > 0; off >>= 1) cnt[q] += __shfl_xor(cnt[q], off);
    if (lane == 0 && cnt[q] != 0) atomicAdd(&counters[4 * (cb % PP_CSLOTS) + q], cnt[q]);   // (slots: same-address atomics serialise, ~12 ns each)
  }
}

// Schur tile: half of an 8x8 tile (8 rows x 4 columns, blockIdx.z selects the column half) in
// registers over all panels holding rows of both tile ranges, then summed over the 64 instances of
// the wave through LDS.  Two waves per tile halve the register footprint (4 waves/SIMD).
// The workgroups in front of the ntile_all * nchunk tile workgroups (z = 0 only) count the inertia codes and collect the
// growth flags (the work of k_count_codes) beside the tiles instead of in a launch of their own in front of them.
__global__ __launch_bounds__(64) void k_schur_tiles(GroupDev g, int ntile_all, size_t total8, int* counters) {
  __shared__ double red[32][65];
  const int lane = threadIdx.x;
  const unsigned ncb = gridDim.x - (unsigned)(ntile_all * g.nchunk);    // counting workgroups come first in the grid
  if (blockIdx.x < ncb) {
    if (blockIdx.z != 0) return;
    count_codes_block(g, blockIdx.x, ncb, total8, counters, lane);
    return;
  }
  const unsigned wg = blockIdx.x - ncb;
  const int chunk = pp_chunk64_perm(wg % (unsigned)g.nchunk, (unsigned)g.nchunk);
  const int b = chunk * 64 + lane;
  const size_t bpad = (size_t)g.bpad;
  const int tile = (int)(wg / (unsigned)g.nchunk), half = blockIdx.z;
  double acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  // column-step records {w, position of (row 0, column t) of the panel, -, -, slotA[8], slotB[8]}: one per panel
  // column holding rows of both tile ranges; four steps (48 loads) are in flight together
  const int r0 = g.stile_ptr[tile], r1 = g.stile_ptr[tile + 1];
  constexpr int SG = 4;
  for (int r = r0; r < r1; r += SG) {
    double la[SG][8], ub[SG][4];
#pragma unroll
    for (int s = 0; s < SG; ++s) {
      const bool live = r + s < r1;
      const int* rec = g.stile_rec + 20 * (size_t)min(r + s, r1 - 1);
      const int w = rec[0];
      const double* Up = g.U + (size_t)rec[1] * bpad + b;
      const double* Lp = g.L + (size_t)rec[1] * bpad + b;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int sa = rec[4 + i];
        const double v = Lp[(size_t)(max(sa, 0) * w) * bpad];
        la[s][i] = (sa >= 0 && live) ? v : 0.0;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int sb = rec[12 + 4 * half + j];
        const double v = Up[(size_t)(max(sb, 0) * w) * bpad];
        ub[s][j] = (sb >= 0) ? v : 0.0;
      }
    }
#pragma unroll
    for (int s = 0; s < SG; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] -= la[s][i] * ub[s][j];
  }
  if (g.cmapT) {
    // mapped group (time blocks: every instance has coupling rows of its own): no sum over the lanes, the clique of
    // every instance is kept and scattered by k_scatter_schur
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        g.Sloc[((size_t)tile * 64 + i * 8 + 4 * half + j) * bpad + b] = acc[i][j];
    return;
  }
  const double mask = (b < g.batch) ? 1.0 : 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[i * 4 + j][lane] = acc[i][j] * mask;
  __syncthreads();
  if (lane < 32) {
    double s = 0.0;
    for (int l = 0; l < 64; ++l) s += red[lane][l];
    // entry (i, 4*half + j) of the tile -> slot i*8 + 4*half + j of the 64-entry tile record
    const int i = lane >> 2, j = lane & 3;
    g.Spart[((size_t)chunk * ntile_all + tile) * 64 + i * 8 + 4 * half + j] = s;
  }
}

// S[ci][cj] += sum over chunks of the tile partials (both triangles of the dense S)
// overwrite: the tiles of this group cover all of S and it is the first group: S = instead of S += (no memset in front)
__global__ __launch_bounds__(64) void k_schur_reduce(GroupDev g, int ntiles, double* __restrict__ S,
                                                     int* __restrict__ counters, int overwrite) {
  const int lane = threadIdx.x, tile = blockIdx.x;
  // last group of the handle: the inertia counters (complete: the counting workgroups ran in the launch before this
  // one) go to the tail of the S buffer, so that they travel with the all-reduce, and are cleared for the next
  // factorisation (saves the one-thread k_write_tail launch and a memset)
  if (counters && tile == 0) {
    int4 c = reinterpret_cast<int4*>(counters)[lane];       // PP_CSLOTS == 64: one slot per lane
    reinterpret_cast<int4*>(counters)[lane] = make_int4(0, 0, 0, 0);
    for (int off = 32; off > 0; off >>= 1) {
      c.x += __shfl_xor(c.x, off); c.y += __shfl_xor(c.y, off); c.z += __shfl_xor(c.z, off); c.w += __shfl_xor(c.w, off);
    }
    if (lane == 0) {
      double* tail = S + (size_t)g.nc * g.nc;
      tail[0] = (double)c.z;
      tail[1] = (double)c.x;
      tail[2] = (double)c.y;
      tail[3] = 0.0;
      tail[4] = (double)c.w;
      tail[5] = tail[6] = tail[7] = 0.0;
    }
  }
  double s = 0.0;
  for (int c = 0; c < g.nchunk; ++c) s += g.Spart[((size_t)c * ntiles + tile) * 64 + lane];
  const int ci = g.stile_a[tile] * 8 + (lane >> 3), cj = g.stile_b[tile] * 8 + (lane & 7);
  if (ci < g.nc && cj < g.nc && ci >= cj) {
    const size_t lo = (size_t)ci + (size_t)cj * g.nc, up = (size_t)cj + (size_t)ci * g.nc;
    if (overwrite) { S[lo] = s; if (ci != cj) S[up] = s; }
    else { S[lo] += s; if (ci != cj) S[up] += s; }
  }
}

// MFMA form of the Schur update for unmapped groups (round 2).  S = - sum over panel columns and INSTANCES of
// (coupling part of the L column)(coupling part of the U column)^T: the instance index is a genuine GEMM K dimension
// -- every operand row is a contiguous [instance] vector -- so a 16 x 16 tile of S over one panel column and the 64
// instances of a chunk is 16 v_mfma_f64_16x16x4 with K = 4 instances each.  Lane (li, lk) feeds row li of the tile's row
// (A, from L) and column (B, from U) ranges with the instances 16 lk .. 16 lk + 15 of the chunk (any assignment of
// instances to K slots is fine as long as A and B agree): 128 contiguous bytes per lane and operand, eight 16-byte
// loads, and a wave reads each 512-byte row of the chunk exactly once.  Against the register-tile kernel (k_schur_tiles:
// 12 loads of 8 bytes per lane for 32 multiply-adds) that is a third of the operand traffic per multiply-add and a
// tenth of the instructions.  Records: per (tile, panel column) the 16 + 16 row positions, -1 where the panel has no row.
// nsplit waves share the records of a tile; partial tiles go to Spart [chunk][tile][split][256] and are added in fixed order
// by k_schur_reduce_mfma (deterministic, no atomics).
__global__ __launch_bounds__(64) void k_schur_mfma(GroupDev g, int nwork_items, size_t total8, int* counters) {
  const int lane = threadIdx.x;
  const unsigned nwork = (unsigned)(nwork_items * g.nchunk);
  const unsigned ncb = gridDim.x - nwork;                                // counting workgroups come first in the grid
  if (blockIdx.x < ncb) { count_codes_block(g, blockIdx.x, ncb, total8, counters, lane); return; }
  const unsigned wg = blockIdx.x - ncb;
  const int chunk = pp_chunk64_perm(wg % (unsigned)g.nchunk, (unsigned)g.nchunk);
  const int item = (int)(wg / (unsigned)g.nchunk);       // a slice of at most PP_MT_SLICE records of one tile
  const int li = lane & 15, lk = lane >> 4;
  const size_t bpad = (size_t)g.bpad;
  const int ra = g.mt_item[2 * item], rb = g.mt_item[2 * item + 1];
  const size_t lane_off = (size_t)chunk * 64 + (size_t)lk * 16;      // first of this lane's 16 instances
  // instances beyond the batch (ragged last chunk) hold undefined factor values: their K slots are zeroed
  const int nvalid = min(16, max(0, g.batch - (int)lane_off));
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  for (int r = ra; r < rb; r += 2) {
    // two records at a time: all their operands are requested before the first multiply
    int oa[2], ob[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bool live = r + s < rb;
      oa[s] = live ? g.mt_rec[(size_t)(r + s) * 32 + li] : -1;
      ob[s] = live ? g.mt_rec[(size_t)(r + s) * 32 + 16 + li] : -1;
    }
    double2 a[2][8], b[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { a[s][q] = make_double2(0.0, 0.0); b[s][q] = make_double2(0.0, 0.0); }
      if (oa[s] >= 0) {         // (rows the panel does not have are not requested)
        const double2* pa = reinterpret_cast<const double2*>(g.L + (size_t)oa[s] * bpad + lane_off);
#pragma unroll
        for (int q = 0; q < 8; ++q) a[s][q] = pa[q];
      }
      if (ob[s] >= 0) {
        const double2* pb = reinterpret_cast<const double2*>(g.U + (size_t)ob[s] * bpad + lane_off);
#pragma unroll
        for (int q = 0; q < 8; ++q) b[s][q] = pb[q];
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const double a0 = (2 * q < nvalid) ? a[s][q].x : 0.0, a1 = (2 * q + 1 < nvalid) ? a[s][q].y : 0.0;
        const double b0 = (2 * q < nvalid) ? b[s][q].x : 0.0, b1 = (2 * q + 1 < nvalid) ? b[s][q].y : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
      }
    }
  }
  // D[row = lk + 4 r][col = li] (see k_ldl_regs): slot (row * 16 + col) of the 256-entry partial tile
  double* out = g.Spart + ((size_t)chunk * nwork_items + item) * 256;
#pragma unroll
  for (int r = 0; r < 4; ++r) out[(lk + 4 * r) * 16 + li] = -acc[r];
}

// The same with 32 x 32 super-tiles (2 x 2 tiles per wave) for large coupling dimensions (n_c >= PP_MT_WIDE_NC: the
// 1000 x 1000 S of C5 has 2016 tiles).  A 16 x 16 tile reads 16 + 16 operand rows for 16 matrix instructions per panel
// column and chunk -- 2 flop per byte, HBM/L2-bound at ~10 TFLOP/s (C5: 23 % MFMA-busy); a super-tile reads 32 + 32 rows
// for 64.  Records: 64 row positions (A rows of the two row tiles, B rows of the two column tiles); work items
// {first record, end, super-tile, -}; partial tiles [chunk][item][2 a + b][256].  With the 91 tiles of n_c = 200 the
// super-tiles were slower (too few waves), hence the threshold.
constexpr int PP_MT_WIDE_NC = 512;
__global__ __launch_bounds__(64) void k_schur_mfma_wide(GroupDev g, int nwork_items, size_t total8, int* counters) {
  const int lane = threadIdx.x;
  const unsigned nwork = (unsigned)(nwork_items * g.nchunk);
  const unsigned ncb = gridDim.x - nwork;                                // counting workgroups come first in the grid
  if (blockIdx.x < ncb) { count_codes_block(g, blockIdx.x, ncb, total8, counters, lane); return; }
  const unsigned wg = blockIdx.x - ncb;
  const int chunk = pp_chunk64_perm(wg % (unsigned)g.nchunk, (unsigned)g.nchunk);
  const int item = (int)(wg / (unsigned)g.nchunk);
  const int li = lane & 15, lk = lane >> 4;
  const size_t bpad = (size_t)g.bpad;
  const int ra = g.mt_item[4 * item], rb = g.mt_item[4 * item + 1], super = g.mt_item[4 * item + 2];
  const bool upper_used = g.mt_a[4 * super + 1] >= 0;                 // (false on the diagonal: tile (2 s, 2 s + 1) lies above it)
  const size_t lane_off = (size_t)chunk * 64 + (size_t)lk * 16;
  const int nvalid = min(16, max(0, g.batch - (int)lane_off));
  double4_t acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = double4_t{0.0, 0.0, 0.0, 0.0};
  for (int r = ra; r < rb; ++r) {
    int oa[2], ob[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      oa[x] = g.mt_rec[(size_t)r * 64 + 16 * x + li];
      ob[x] = g.mt_rec[(size_t)r * 64 + 32 + 16 * x + li];
    }
    double2 a[2][8], b[2][8];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { a[x][q] = make_double2(0.0, 0.0); b[x][q] = make_double2(0.0, 0.0); }
      if (oa[x] >= 0) {
        const double2* pa = reinterpret_cast<const double2*>(g.L + (size_t)oa[x] * bpad + lane_off);
#pragma unroll
        for (int q = 0; q < 8; ++q) a[x][q] = pa[q];
      }
      if (ob[x] >= 0) {
        const double2* pb = reinterpret_cast<const double2*>(g.U + (size_t)ob[x] * bpad + lane_off);
#pragma unroll
        for (int q = 0; q < 8; ++q) b[x][q] = pb[q];
      }
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (!(2 * q < nvalid)) { a[x][q].x = 0.0; b[x][q].x = 0.0; }
        if (!(2 * q + 1 < nvalid)) { a[x][q].y = 0.0; b[x][q].y = 0.0; }
      }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        if (x == 0 && y == 1 && !upper_used) continue;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x][q].x, b[y][q].x, acc[x][y], 0, 0, 0);
          acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x][q].y, b[y][q].y, acc[x][y], 0, 0, 0);
        }
      }
  }
  double* out = g.Spart + ((size_t)chunk * nwork_items + item) * 1024;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(2 * x + y) * 256 + (lk + 4 * r) * 16 + li] = -acc[x][y][r];
}

// S[ci][cj] (+)= sum over the work items of the tile and the chunks of the partial 16 x 16 tiles (both triangles of the dense
// S).  Sixteen partial sums per entry (one per residue of the chunk index; fixed order inside: deterministic) meet in LDS
// and are added as a fixed tree.  Tail as in k_schur_reduce.
__global__ __launch_bounds__(1024) void k_schur_reduce_mfma(GroupDev g, int nwork_items, double* __restrict__ S,
                                                            int* __restrict__ counters, int overwrite, int wide) {
  __shared__ double part[4][256];
  const int tid = threadIdx.x, tile = blockIdx.x, e = tid & 255, sub = tid >> 8;
  if (counters && tile == 0 && tid < 64) {
    const int lane = tid;
    int4 c = reinterpret_cast<int4*>(counters)[lane];       // PP_CSLOTS == 64: one slot per lane
    reinterpret_cast<int4*>(counters)[lane] = make_int4(0, 0, 0, 0);
    for (int off = 32; off > 0; off >>= 1) {
      c.x += __shfl_xor(c.x, off); c.y += __shfl_xor(c.y, off); c.z += __shfl_xor(c.z, off); c.w += __shfl_xor(c.w, off);
    }
    if (lane == 0) {
      double* tail = S + (size_t)g.nc * g.nc;
      tail[0] = (double)c.z;
      tail[1] = (double)c.x;
      tail[2] = (double)c.y;
      tail[3] = 0.0;
      tail[4] = (double)c.w;
      tail[5] = tail[6] = tail[7] = 0.0;
    }
  }
  if (g.mt_a[tile] < 0) return;          // (wide form: the unused quarter of a diagonal super-tile; uniform per workgroup, before any barrier)
  // wide form: tile = 4 * super-tile + quarter, the items belong to the super-tile, partial tiles [chunk][item][quarter][256]
  const int i0 = g.mt_wptr[wide ? tile >> 2 : tile], i1 = g.mt_wptr[(wide ? tile >> 2 : tile) + 1];
  const size_t istride = wide ? 1024 : 256, qoff = wide ? (size_t)(tile & 3) * 256 : 0;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int c = sub; c < g.nchunk; c += 4) {                  // (the four quarter-sums: chunks 0,4,8,.. / 1,5,9,.. / ...)
    const double* base = g.Spart + ((size_t)c * nwork_items) * istride + qoff + e;
    int it = i0;
    for (; it + 3 < i1; it += 4) {
      s0 += base[(size_t)it * istride]; s1 += base[(size_t)(it + 1) * istride]; s2 += base[(size_t)(it + 2) * istride]; s3 += base[(size_t)(it + 3) * istride];
    }
    for (; it < i1; ++it) s0 += base[(size_t)it * istride];
  }
  part[sub][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sub != 0) return;
  const double s = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
  const int ci = g.mt_a[tile] * 16 + (e >> 4), cj = g.mt_b[tile] * 16 + (e & 15);
  if (ci < g.nc && cj < g.nc && ci >= cj) {
    const size_t lo = (size_t)ci + (size_t)cj * g.nc, up = (size_t)cj + (size_t)ci * g.nc;
    if (overwrite) { S[lo] = s; if (ci != cj) S[up] = s; }
    else { S[lo] += s; if (ci != cj) S[up] += s; }
  }
}

// Mapped groups: S[gi][gj] += clique entry (ci, cj) of instance b, gi = cmap[b][ci].  Target: the dense n_c x n_c S
// (both triangles) or the block-tridiagonal storage D[G][gs][gs] | E[G-1][gs][gs] (E_t = rows of block t+1 x columns of
// block t, column-major inside a block).  Entries of different instances may coincide in general: atomic adds.
struct SchurTarget { double* S; int nc, btd, gs, G; int* err; };

__device__ __forceinline__ void schur_add(const SchurTarget& T, int gi, int gj, double v) {
  if (!T.btd) {
    atomicAdd(&T.S[(size_t)gi + (size_t)gj * T.nc], v);
    if (gi != gj) atomicAdd(&T.S[(size_t)gj + (size_t)gi * T.nc], v);
    return;
  }
  const int bi = gi / T.gs, bj = gj / T.gs, ri = gi % T.gs, rj = gj % T.gs;
  const size_t g2 = (size_t)T.gs * T.gs;
  if (bi == bj) {
    atomicAdd(&T.S[(size_t)bi * g2 + ri + (size_t)rj * T.gs], v);
    if (gi != gj) atomicAdd(&T.S[(size_t)bi * g2 + rj + (size_t)ri * T.gs], v);
  } else if (bi == bj + 1) {
    atomicAdd(&T.S[(size_t)T.G * g2 + (size_t)bj * g2 + ri + (size_t)rj * T.gs], v);
  } else if (bj == bi + 1) {
    atomicAdd(&T.S[(size_t)T.G * g2 + (size_t)bi * g2 + rj + (size_t)ri * T.gs], v);
  } else {
    T.err[0] = 1;       // a clique that spans non-adjacent blocks: the structure given to pp_set_coupling_structure is wrong
  }
}

__global__ __launch_bounds__(64) void k_scatter_schur(GroupDev g, int ntiles, SchurTarget T) {
  const int lane = threadIdx.x;
  const int tile = PP_TASK_OF_WG(g.nchunk), b = PP_CHUNK_OF_WG(g.nchunk) * 64 + lane;
  if (b >= g.batch) return;
  const size_t bpad = (size_t)g.bpad;
  const int ta = g.stile_a[tile], tb = g.stile_b[tile];
  for (int e = 0; e < 64; ++e) {
    const int ci = ta * 8 + (e >> 3), cj = tb * 8 + (e & 7);
    if (ci >= g.nc || cj >= g.nc || ci < cj) continue;
    const double v = g.Sloc[((size_t)tile * 64 + e) * bpad + b];
    if (v == 0.0) continue;
    schur_add(T, g.cmapT[(size_t)ci * bpad + b], g.cmapT[(size_t)cj * bpad + b], v);
  }
}

// per-instance copy of the coupling solution for the back substitution of a mapped group
__global__ __launch_bounds__(256) void k_gather_xc(GroupDev g, const double* __restrict__ xc) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)g.nc * g.bpad) return;
  const int b = (int)(i % g.bpad);
  g.XCL[i] = (b < g.batch) ? xc[g.cmapT[i]] : 0.0;
}

__global__ void k_write_tail(int* counters, double* tail) {
  if (threadIdx.x == 0) {
    int c[4] = {0, 0, 0, 0};
    for (int sl = 0; sl < PP_CSLOTS; ++sl)
      for (int q = 0; q < 4; ++q) { c[q] += counters[4 * sl + q]; counters[4 * sl + q] = 0; }   // (ready for the next factorisation)
    tail[0] = (double)c[2];  // numerically zero pivots
    tail[1] = (double)c[0];
    tail[2] = (double)c[1];
    tail[3] = 0.0;
    tail[4] = (double)c[3];
    tail[5] = tail[6] = tail[7] = 0.0;
  }
}

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

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's outstanding GLOBAL stores
// (release fence at workgroup scope): in a loop that streams finished columns to global memory and never reads them
// back, that puts one global-store round trip (1-2 us) on the critical path of every barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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

template <bool DPP>
__global__ __launch_bounds__(LDL_THREADS) void k_ldl_regs(int n, const double* __restrict__ S, const double* __restrict__ Q,
                                                          double* __restrict__ A, double* __restrict__ dvec,
                                                          int* __restrict__ mode, int* __restrict__ info, double eps) {
  __shared__ double P[16 * LDLR_NT][LDLR_LD];
  __shared__ double dl[LDLR_NB], rdl[LDLR_NB];
  __shared__ double red[LDL_THREADS / 64];
  __shared__ int sflags[2];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = LDL_THREADS / 64;
  const int li = lane & 15, lk = lane >> 4;
  const size_t lda = (size_t)n;
  const int nt = (n + 15) / 16, ntt = nt * (nt + 1) / 2;
  // the input is S + Q (Q: lower triangle authoritative, may be null), read straight from the all-reduced buffer:
  // no separate add/copy kernel in front of the factorisation; S itself stays untouched for the pivoted fallback
  double loc = 0.0;
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
        v = S[col + (size_t)row * lda];
        if (Q) v += (row >= col) ? Q[row + (size_t)col * lda] : Q[col + (size_t)row * lda];
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
    const bool ok = (sflags[0] == 0) && (sflags[1] == 1 || sflags[1] == 2 || n == 0);
    mode[0] = ok ? 1 : 0;
    if (ok) { info[0] = (sflags[1] == 1) ? n : 0; info[1] = (sflags[1] == 2) ? n : 0; info[2] = 0; }
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

// ------------------------------------------------------------------------------------------
// Block-tridiagonal S (time-staged problems; the reference factorises a sparse COO S with its sub-solver,
// mpi_...:88-125, 228-255, 352-361).  Storage: D[G][gs][gs] | E[G-1][gs][gs], E_t = S(block t+1, block t), column-major
// inside a block.  Factorised by BLOCK CYCLIC REDUCTION: at level l (stride s = 2^l) the blocks i = s (2k + 1) are
// eliminated together -- Bunch-Kaufman of D_i, explicit inverse (gs unit right-hand sides), Y_lo = inv_i S(i, i-s),
// Y_up = inv_i S(i, i+s), then D_{i-s} -= S(i-s, i) Y_lo, D_{i+s} -= S(i+s, i) Y_up and the new coupling
// S(i+s, i-s) = -S(i+s, i) Y_lo -- log2 G levels of batched block operations instead of G dependent steps; block 0 is
// eliminated last.  inertia(S) = sum of the inertias of the eliminated D_i (Haynsworth).  The solve walks the same
// levels with matrix-vector products only.  slot j of the coupling array holds S(j + s, j) at the current level; the
// couplings of an eliminated block are copied to Klo / Kup for the solve.
struct BcrLevel { const int* elim; int ne, s, lo; };   // lo = 0: the lower neighbour i - s is already eliminated (sequential order)

// F[pos[k]] += val[k]: the coupling block Q of a time-staged problem (sc_ip_interface.py:308-357: -I couplings between
// link duals and coupling states, regularisation on the diagonal) has a few entries per row -- handed over as
// (position, value) pairs in the layout of the Schur buffer instead of a flat array of that (tens of MB) size
__global__ __launch_bounds__(256) void k_corner_add(long long nnz, const long long* __restrict__ pos, const double* __restrict__ val,
                                                    double* __restrict__ F) {
  const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
  if (k < nnz) atomicAdd(&F[pos[k]], val[k]);
}

__global__ __launch_bounds__(256) void k_btd_init(size_t n, const double* __restrict__ S, const double* __restrict__ Q,
                                                  double* __restrict__ F) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) F[i] = S[i] + (Q ? Q[i] : 0.0);
}

// in_lds: the block is factorised in LDS (gs * gs doubles of dynamic shared memory: every one of the gs pivot steps is
// a handful of LDS round trips instead of global-memory ones) and copied back
__global__ __launch_bounds__(BK_THREADS) void k_bcr_factor(int gs, BcrLevel lv, double* D, int* ipiv, double* work, int* info,
                                                           int in_lds, int skip_accepted) {
  extern __shared__ __attribute__((aligned(16))) double shD[];
  __shared__ double sv[16];
  __shared__ int si[16];
  __shared__ pp::BkInfo sbi;
  const int i = lv.elim[blockIdx.x];
  if (skip_accepted && info[4 * i + 3] == 1) return;      // inverted by k_bcr_ldl_inverse
  double* Dg = D + (size_t)i * gs * gs;
  double* A = Dg;
  double* wk = work + (size_t)blockIdx.x * 2 * gs;
  if (in_lds) {
    // block, the two work columns of a 2 x 2 pivot step and the pivot indices all live in LDS, and the barriers of the
    // factorisation order LDS traffic only: no global-memory round trip inside the gs pivot steps
    for (int k = threadIdx.x; k < gs * gs; k += blockDim.x) shD[k] = Dg[k];
    __syncthreads();
    int* lpiv = reinterpret_cast<int*>(shD + (size_t)gs * gs + 2 * (size_t)gs);
    TeamCtxLds ctx{sv, si};
    pp::bk_factor(ctx, gs, shD, gs, lpiv, shD + (size_t)gs * gs, &sbi, BK_EPS);
    __syncthreads();
    for (int k = threadIdx.x; k < gs * gs; k += blockDim.x) Dg[k] = shD[k];
    for (int k = threadIdx.x; k < gs; k += blockDim.x) ipiv[(size_t)i * gs + k] = lpiv[k];
  } else {
    TeamCtx ctx{sv, si};
    pp::bk_factor(ctx, gs, A, gs, ipiv + (size_t)i * gs, wk, &sbi, BK_EPS);
  }
  if (threadIdx.x == 0) { info[4 * i] = sbi.npos; info[4 * i + 1] = sbi.nneg; info[4 * i + 2] = sbi.nzero; info[4 * i + 3] = 0; }
}

// column j of inv(D_i): Bunch-Kaufman solve of the unit vector e_j (one workgroup per column and block)
__global__ __launch_bounds__(128) void k_bcr_invert(int gs, BcrLevel lv, const double* __restrict__ D, const int* __restrict__ ipiv,
                                                    double* __restrict__ inv) {
  __shared__ double sv[16];
  __shared__ int si[16];
  const int i = lv.elim[blockIdx.y];
  double* col = inv + (size_t)i * gs * gs + (size_t)blockIdx.x * gs;
  for (int r = threadIdx.x; r < gs; r += blockDim.x) col[r] = (r == (int)blockIdx.x) ? 1.0 : 0.0;
  __syncthreads();
  TeamCtx ctx{sv, si};
  pp::bk_solve(ctx, gs, D + (size_t)i * gs * gs, gs, ipiv + (size_t)i * gs, col);
}

// The same, one WAVE per column (four columns per workgroup), the vector in LDS: the ~2 gs team synchronisations of a
// solve are wave-level (free) instead of workgroup barriers.  Dynamic LDS: 4 * gs doubles.  (MEASURED at C4, gs = 98:
// 350 -> 202 us per level; with the factored block in LDS as well -- 16 columns per workgroup, 89 KB -- 220 us: a step is a
// chain of dependent accesses either way, and the smaller footprint keeps twice as many waves on a CU.)
__global__ __launch_bounds__(256) void k_bcr_invert_wave(int gs, BcrLevel lv, const double* __restrict__ D, const int* __restrict__ ipiv,
                                                         double* __restrict__ inv, const int* __restrict__ accepted) {
  extern __shared__ __attribute__((aligned(16))) double shv[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lv.elim[blockIdx.y], c = (int)blockIdx.x * 4 + wave;
  if (c >= gs) return;
  if (accepted && accepted[4 * i + 3] == 1) return;       // inverted by k_bcr_ldl_inverse
  double* b = shv + (size_t)wave * gs;
  for (int r = lane; r < gs; r += 64) b[r] = (r == c) ? 1.0 : 0.0;
  WaveCtx ctx;
  ctx.sync();
  pp::bk_solve(ctx, gs, D + (size_t)i * gs * gs, gs, ipiv + (size_t)i * gs, b);
  double* col = inv + (size_t)i * gs * gs + (size_t)c * gs;
  for (int r = lane; r < gs; r += 64) col[r] = b[r];
}

// Fast path of the two kernels above for gs <= 16 * BL_NT: UNPIVOTED blocked LDL^T of D_i on the matrix cores with the
// whole block in LDS, followed by the explicit inverse from that factor -- one launch per level instead of a
// Bunch-Kaufman factorisation (gs pivot searches, each a workgroup reduction) and gs wave-level solves.
//   acceptance (else the block is left to k_bcr_factor / k_bcr_invert_wave, which skip accepted blocks): no pivot below
//   BK_EPS * max|diagonal| and no multiplier above `lbound` in magnitude -- the 1 x 1 pivots then satisfy the threshold
//   test |d_k| >= max_i |a_ik| / lbound of the reference's sub-solver (MA27's u, ma27_interface.py:36-47); the blocks of
//   a time-staged S are quasi-definite ([-P1 *; * P2], link duals and coupling states), which an LDL^T without
//   interchanges factorises for any ordering.  Pivot signs give the inertia (Haynsworth, as before).
//   tiles: 16 x 16, row stride BL_LD, tile (I, J), I >= J, at (I (I + 1) / 2 + J) * BL_TILE; T holds A then L, Z = inv(L)
//   phase 1  right-looking LDL^T: diagonal tile in the registers of wave 0 (DiagSteps), panel rows one per thread
//            (PanelSolveCols), trailing tiles T(I,J) -= (L_Ip D_p) L_Jp^T by fp64 MFMA, operands and result in LDS
//   phase 2  Z = inv(L): diagonal tiles by column substitution (lane = column), then by block diagonals t = I - J:
//            Z_IJ = -Z_II sum_{J <= K < I} L_IK Z_KJ  (the MFMA result layout of the sum IS the operand layout of the second product)
//   phase 3  inv(D_i) = Z^T D^-1 Z, tile (I, J) = sum_{K >= I} Z_KI^T D_K^-1 Z_KJ, written to both triangles
constexpr int BL_NT = 7;
constexpr int BL_LD = 18;
constexpr int BL_TILE = 16 * BL_LD;
constexpr int BL_NTT = BL_NT * (BL_NT + 1) / 2;
constexpr int BL_THREADS = 512;
constexpr size_t BL_LDS_BYTES = 2 * (size_t)BL_NTT * BL_TILE * sizeof(double);

// C[i][j] += sum_k a(i, k) b(k, j) over one 16-wide k block: av[q] = a(li, 4 q + lk), bv[q] = b(4 q + lk, li);
// the result register r of lane (li, lk) is C[lk + 4 r][li]
__device__ __forceinline__ double4_t bl_mma(const double (&av)[4], const double (&bv)[4], double4_t acc) {
#pragma unroll
  for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc, 0, 0, 0);
  return acc;
}

__global__ __launch_bounds__(BL_THREADS) void k_bcr_ldl_inverse(int gs, BcrLevel lv, const double* __restrict__ D,
                                                                double* __restrict__ inv, int* __restrict__ info, double eps,
                                                                double lbound) {
  extern __shared__ __attribute__((aligned(16))) double blsh[];
  double* T = blsh;
  double* Z = blsh + (size_t)BL_NTT * BL_TILE;
  __shared__ double dl[16 * BL_NT], rdl[16 * BL_NT], dmag[16 * BL_NT];
  __shared__ double red[BL_THREADS / 64];
  __shared__ int sflags[2], orig[16 * BL_NT];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = BL_THREADS / 64;
  const int li = lane & 15, lk = lane >> 4;
  const int i = lv.elim[blockIdx.x];
  const double* Dg = D + (size_t)i * gs * gs;
  const int nt = (gs + 15) / 16, ntt = nt * (nt + 1) / 2;
  // ---- symmetric pre-ordering by decreasing |diagonal| (static, from the values of this factorisation: the coupling
  // states of a time-staged S carry O(1) diagonals, the link duals nearly none -- states first makes every pivot of
  // the quasi-definite block its column's largest entry; the acceptance test below judges the result)
  for (int k = tid; k < 16 * BL_NT; k += BL_THREADS) {
    dmag[k] = (k < gs) ? fabs(Dg[(size_t)k + (size_t)k * gs]) : -1.0;
    orig[k] = k;          // (a NaN diagonal leaves ranks unassigned: the identity keeps every index valid, the pivot test rejects)
  }
  if (tid == 0) { sflags[0] = 0; sflags[1] = 0; }
  __syncthreads();
  double loc = 0.0;
  for (int k = tid; k < 16 * BL_NT; k += BL_THREADS) {
    int rank = k;
    if (k < gs) {
      const double mk = dmag[k];
      rank = 0;
      for (int j = 0; j < gs; ++j) rank += (dmag[j] > mk || (dmag[j] == mk && j < k)) ? 1 : 0;
      loc = mk;
    }
    orig[rank] = k;
  }
  for (int off = 32; off > 0; off >>= 1) loc = fmax(loc, __shfl_xor(loc, off));
  if (lane == 0) red[wv] = loc;
  __syncthreads();
  double anorm = 0.0;
  for (int q = 0; q < nwv; ++q) anorm = fmax(anorm, red[q]);
  // ---- load: lower tiles (diagonal tiles in full, mirrored from the lower triangle), identity padding
  for (int idx = tid; idx < ntt * 256; idx += BL_THREADS) {
    const int t = idx >> 8, e = idx & 255, r = e & 15, c = e >> 4;
    int I = 0;
    while ((I + 1) * (I + 2) / 2 <= t) ++I;
    const int J = t - I * (I + 1) / 2;
    const int gr = 16 * I + r, gc = 16 * J + c;
    double v = (gr == gc) ? 1.0 : 0.0;
    if (gr < gs && gc < gs) {
      const int o_r = orig[gr], o_c = orig[gc];
      v = Dg[(size_t)max(o_r, o_c) + (size_t)min(o_r, o_c) * gs];
    }
    T[(size_t)t * BL_TILE + r * BL_LD + c] = v;
  }
  __syncthreads();
  // ---- phase 1
  for (int p = 0; p < nt; ++p) {
    double* Tpp = T + (size_t)(p * (p + 1) / 2 + p) * BL_TILE;
    const int nb = min(16, gs - 16 * p), m = 16 * (nt - p - 1);
    if (wv == 0) {
      double row[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) row[j] = Tpp[li * BL_LD + j];
      int bad = 0, signs = 0;
      double d = mov_row_bcast<0>(row[0]);
      if (!(fabs(d) > eps * anorm)) { bad = 1; d = (anorm > 0.0 ? anorm : 1.0); }
      double rd = fast_rcp(d);
      DiagSteps<0>::run(row, d, rd, bad, signs, nb, eps * anorm, anorm, dl + 16 * p, rdl + 16 * p, lane);
      bool big = false;
#pragma unroll
      for (int j = 0; j < 16; ++j) big = big || (j < li && !(fabs(row[j]) <= lbound));
      if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) Tpp[lane * BL_LD + j] = row[j];
      }
      if (bad || big) sflags[0] = 1;
    }
    lds_barrier();
    if (64 * wv < m) {        // W = A21 L11^{-T} (thread = row of the panel), L21 = W D^-1
      const int r = min(tid, m - 1);
      double* Trow = T + (size_t)((p + 1 + (r >> 4)) * (p + 2 + (r >> 4)) / 2 + p) * BL_TILE + (r & 15) * BL_LD;
      double lrow[16], wrow[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) { lrow[k] = Tpp[li * BL_LD + k]; wrow[k] = Trow[k]; }
      PanelSolveCols<0>::run(wrow, lrow);
      bool big = false;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        wrow[k] *= rdl[16 * p + k];
        big = big || !(fabs(wrow[k]) <= lbound);
      }
      if (tid < m) {
#pragma unroll
        for (int k = 0; k < 16; ++k) Trow[k] = wrow[k];
        if (big) sflags[0] = 1;
      }
    }
    lds_barrier();
    const int q1 = nt - p - 1, cnt = q1 * (q1 + 1) / 2;
    for (int idx = wv; idx < cnt; idx += nwv) {
      int Ir = 0;
      while ((Ir + 1) * (Ir + 2) / 2 <= idx) ++Ir;
      const int I = p + 1 + Ir, J = p + 1 + idx - Ir * (Ir + 1) / 2;
      const double* TIp = T + (size_t)(I * (I + 1) / 2 + p) * BL_TILE;
      const double* TJp = T + (size_t)(J * (J + 1) / 2 + p) * BL_TILE;
      double* TIJ = T + (size_t)(I * (I + 1) / 2 + J) * BL_TILE;
      double av[4], bv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = 4 * q + lk;
        av[q] = -TIp[li * BL_LD + k] * dl[16 * p + k];
        bv[q] = TJp[li * BL_LD + k];
      }
      double4_t acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = TIJ[(lk + 4 * r) * BL_LD + li];
      acc = bl_mma(av, bv, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) TIJ[(lk + 4 * r) * BL_LD + li] = acc[r];
    }
    lds_barrier();
  }
  const bool ok = sflags[0] == 0;
  if (!ok) {                                   // left to the pivoted kernels
    if (tid == 0) info[4 * i + 3] = 0;
    return;
  }
  // ---- phase 2: diagonal tiles of Z
  if (wv < nt) {
    const double* Lt = T + (size_t)(wv * (wv + 1) / 2 + wv) * BL_TILE;
    double z[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = (r == li) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < 15; ++k) {
#pragma unroll
      for (int r = k + 1; r < 16; ++r) z[r] -= Lt[r * BL_LD + k] * z[k];
    }
    if (lane < 16) {
      double* Zt = Z + (size_t)(wv * (wv + 1) / 2 + wv) * BL_TILE;
#pragma unroll
      for (int r = 0; r < 16; ++r) Zt[r * BL_LD + lane] = z[r];
    }
  }
  lds_barrier();
  for (int t = 1; t < nt; ++t) {
    const int I = t + wv, J = wv;
    if (I < nt) {
      double4_t R = {0.0, 0.0, 0.0, 0.0};
      for (int K = J; K < I; ++K) {
        const double* TIK = T + (size_t)(I * (I + 1) / 2 + K) * BL_TILE;
        const double* ZKJ = Z + (size_t)(K * (K + 1) / 2 + J) * BL_TILE;
        double av[4], bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          av[q] = TIK[li * BL_LD + 4 * q + lk];
          bv[q] = ZKJ[(4 * q + lk) * BL_LD + li];
        }
        R = bl_mma(av, bv, R);
      }
      const double* ZII = Z + (size_t)(I * (I + 1) / 2 + I) * BL_TILE;
      double av[4], bv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { av[q] = -ZII[li * BL_LD + 4 * q + lk]; bv[q] = R[q]; }
      double4_t zz = {0.0, 0.0, 0.0, 0.0};
      zz = bl_mma(av, bv, zz);
      double* ZIJ = Z + (size_t)(I * (I + 1) / 2 + J) * BL_TILE;
#pragma unroll
      for (int r = 0; r < 4; ++r) ZIJ[(lk + 4 * r) * BL_LD + li] = zz[r];
    }
    lds_barrier();
  }
  // ---- phase 3
  double* X = inv + (size_t)i * gs * gs;
  for (int t = wv; t < ntt; t += nwv) {
    int I = 0;
    while ((I + 1) * (I + 2) / 2 <= t) ++I;
    const int J = t - I * (I + 1) / 2;
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    for (int K = I; K < nt; ++K) {
      const double* ZKI = Z + (size_t)(K * (K + 1) / 2 + I) * BL_TILE;
      const double* ZKJ = Z + (size_t)(K * (K + 1) / 2 + J) * BL_TILE;
      double av[4], bv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = 4 * q + lk;
        av[q] = ZKI[k * BL_LD + li] * rdl[16 * K + k];
        bv[q] = ZKJ[k * BL_LD + li];
      }
      acc = bl_mma(av, bv, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * I + lk + 4 * r, col = 16 * J + li;
      if (row < gs && col < gs) {
        const int o_r = orig[row], o_c = orig[col];
        X[(size_t)o_r + (size_t)o_c * gs] = acc[r];
        if (I != J) X[(size_t)o_c + (size_t)o_r * gs] = acc[r];
      }
    }
  }
  if (tid == 0) {
    int npos = 0, nneg = 0;
    for (int k = 0; k < gs; ++k) { npos += dl[k] > 0.0; nneg += dl[k] < 0.0; }     // (positions < gs are the block's own rows)
    info[4 * i] = npos; info[4 * i + 1] = nneg; info[4 * i + 2] = 0; info[4 * i + 3] = 1;
  }
}

// Klo_i = S(i, i-s) = slot(i-s)^T ... kept as the slot itself: Klo[i] = slot[i-s] (= S(i, i-s), rows of block i);
// Kup[i] = slot[i] (= S(i+s, i)).  Ylo = inv_i Klo,  Yup = inv_i Kup^T.
__global__ __launch_bounds__(256) void k_bcr_keep_y(int gs, int G, BcrLevel lv, const double* __restrict__ inv,
                                                    const double* __restrict__ slot, double* __restrict__ Klo,
                                                    double* __restrict__ Kup, double* __restrict__ Ylo, double* __restrict__ Yup) {
  const int i = lv.elim[blockIdx.y], s = lv.s;
  const size_t g2 = (size_t)gs * gs;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= gs * gs) return;
  const int r = idx % gs, c = idx / gs;
  const double* I = inv + (size_t)i * g2;
  if (lv.lo && i - s >= 0) {
    const double* Sl = slot + (size_t)(i - s) * g2;      // S(i, i-s)
    Klo[(size_t)i * g2 + idx] = Sl[idx];
    double a = 0.0;
    for (int k = 0; k < gs; ++k) a += I[(size_t)r + (size_t)k * gs] * Sl[(size_t)k + (size_t)c * gs];
    Ylo[(size_t)i * g2 + idx] = a;
  }
  if (i + s < G) {
    const double* Su = slot + (size_t)i * g2;            // S(i+s, i)
    Kup[(size_t)i * g2 + idx] = Su[idx];
    double a = 0.0;
    for (int k = 0; k < gs; ++k) a += I[(size_t)r + (size_t)k * gs] * Su[(size_t)c + (size_t)k * gs];   // inv_i S(i, i+s) = inv_i Kup^T
    Yup[(size_t)i * g2 + idx] = a;
  }
}

// which = 0: D_{i-s} -= Klo_i^T Ylo_i   and the new coupling  slot[i-s] = -Kup_i Ylo_i  (if both neighbours exist)
// which = 1: D_{i+s} -= Kup_i Yup_i
__global__ __launch_bounds__(256) void k_bcr_update(int gs, int G, BcrLevel lv, int which, const double* __restrict__ Klo,
                                                    const double* __restrict__ Kup, const double* __restrict__ Ylo,
                                                    const double* __restrict__ Yup, double* __restrict__ D, double* __restrict__ slot) {
  const int i = lv.elim[blockIdx.y], s = lv.s;
  const size_t g2 = (size_t)gs * gs;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= gs * gs) return;
  const int r = idx % gs, c = idx / gs;
  if (which == 0) {
    if (!lv.lo || i - s < 0) return;
    const double* K = Klo + (size_t)i * g2;
    const double* Y = Ylo + (size_t)i * g2;
    double a = 0.0;
    for (int k = 0; k < gs; ++k) a += K[(size_t)k + (size_t)r * gs] * Y[(size_t)k + (size_t)c * gs];      // (Klo^T Ylo)[r][c]
    D[(size_t)(i - s) * g2 + idx] -= a;
    if (i + s < G) {
      const double* U = Kup + (size_t)i * g2;
      double b = 0.0;
      for (int k = 0; k < gs; ++k) b += U[(size_t)r + (size_t)k * gs] * Y[(size_t)k + (size_t)c * gs];    // (Kup Ylo)[r][c]
      slot[(size_t)(i - s) * g2 + idx] = -b;             // S(i+s, i-s) for the next level (stride 2 s)
    }
  } else {
    if (i + s >= G) return;
    const double* U = Kup + (size_t)i * g2;
    const double* Y = Yup + (size_t)i * g2;
    double a = 0.0;
    for (int k = 0; k < gs; ++k) a += U[(size_t)r + (size_t)k * gs] * Y[(size_t)k + (size_t)c * gs];      // (Kup Yup)[r][c]
    D[(size_t)(i + s) * g2 + idx] -= a;
  }
}

// The gs x gs block products of the cyclic reduction on the fp64 matrix cores (round 2): one wave per 16 x 16 tile of
// the product, v_mfma_f64_16x16x4 over K in steps of 4 (column-major blocks, leading dimension gs; TA / TB: the operand is
// the transposed block; entries beyond gs are zeros).  Lane (li, lk) supplies A(m0 + li, k0 + lk) and B(k0 + lk, n0 + li);
// the result lane holds D(m0 + lk + 4 r, n0 + li), r = 0..3 (as in k_ldl_regs).
template <bool TA, bool TB>
__device__ __forceinline__ double4_t bcr_gemm_tile(const double* __restrict__ A, const double* __restrict__ B, int gs, int m0,
                                                   int n0, int li, int lk) {
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  const int m = m0 + li, n = n0 + li;
  for (int k0 = 0; k0 < gs; k0 += 16) {
    double a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {            // four K steps requested together
      const int k = k0 + 4 * u + lk;
      a[u] = (m < gs && k < gs) ? (TA ? A[(size_t)k + (size_t)m * gs] : A[(size_t)m + (size_t)k * gs]) : 0.0;
      b[u] = (n < gs && k < gs) ? (TB ? B[(size_t)n + (size_t)k * gs] : B[(size_t)k + (size_t)n * gs]) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
  }
  return acc;
}

// MFMA form of k_bcr_keep_y: grid (tiles, eliminated blocks, 2): z = 0 the lower neighbour (Klo, Ylo), z = 1 the upper one
__global__ __launch_bounds__(64) void k_bcr_keep_y_mfma(int gs, int G, BcrLevel lv, const double* __restrict__ inv,
                                                        const double* __restrict__ slot, double* __restrict__ Klo,
                                                        double* __restrict__ Kup, double* __restrict__ Ylo, double* __restrict__ Yup) {
  const int i = lv.elim[blockIdx.y], s = lv.s, nt = (gs + 15) / 16;
  const size_t g2 = (size_t)gs * gs;
  const int li = threadIdx.x & 15, lk = threadIdx.x >> 4;
  const int m0 = 16 * (int)(blockIdx.x % nt), n0 = 16 * (int)(blockIdx.x / nt);
  const double* I = inv + (size_t)i * g2;
  if (blockIdx.z == 0) {
    if (!(lv.lo && i - s >= 0)) return;
    const double* Sl = slot + (size_t)(i - s) * g2;      // S(i, i-s)
    const double4_t y = bcr_gemm_tile<false, false>(I, Sl, gs, m0, n0, li, lk);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + lk + 4 * r, col = n0 + li;
      if (row < gs && col < gs) { const size_t idx = (size_t)row + (size_t)col * gs; Ylo[(size_t)i * g2 + idx] = y[r]; Klo[(size_t)i * g2 + idx] = Sl[idx]; }
    }
  } else {
    if (i + s >= G) return;
    const double* Su = slot + (size_t)i * g2;            // S(i+s, i)
    const double4_t y = bcr_gemm_tile<false, true>(I, Su, gs, m0, n0, li, lk);       // inv_i Kup^T
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + lk + 4 * r, col = n0 + li;
      if (row < gs && col < gs) { const size_t idx = (size_t)row + (size_t)col * gs; Yup[(size_t)i * g2 + idx] = y[r]; Kup[(size_t)i * g2 + idx] = Su[idx]; }
    }
  }
}

// MFMA form of k_bcr_update: grid (tiles, eliminated blocks, z); z + zbase = 0: D_{i-s} -= Klo^T Ylo, 1: slot[i-s] = -Kup Ylo,
// 2: D_{i+s} -= Kup Yup (a launch of its own: D_j is updated from both sides within a level).
__global__ __launch_bounds__(64) void k_bcr_update_mfma(int gs, int G, BcrLevel lv, const double* __restrict__ Klo,
                                                        const double* __restrict__ Kup, const double* __restrict__ Ylo,
                                                        const double* __restrict__ Yup, double* __restrict__ D, double* __restrict__ slot,
                                                        int zbase) {
  const int i = lv.elim[blockIdx.y], s = lv.s, nt = (gs + 15) / 16;
  const size_t g2 = (size_t)gs * gs;
  const int li = threadIdx.x & 15, lk = threadIdx.x >> 4;
  const int m0 = 16 * (int)(blockIdx.x % nt), n0 = 16 * (int)(blockIdx.x / nt);
  const int z = (int)blockIdx.z + zbase;
  const bool has_lo = lv.lo && i - s >= 0, has_up = i + s < G;
  double4_t v;
  double* dst;
  double sign;
  bool add;
  if (z == 0) {
    if (!has_lo) return;
    v = bcr_gemm_tile<true, false>(Klo + (size_t)i * g2, Ylo + (size_t)i * g2, gs, m0, n0, li, lk);
    dst = D + (size_t)(i - s) * g2; sign = -1.0; add = true;
  } else if (z == 1) {
    if (!(has_lo && has_up)) return;
    v = bcr_gemm_tile<false, false>(Kup + (size_t)i * g2, Ylo + (size_t)i * g2, gs, m0, n0, li, lk);
    dst = slot + (size_t)(i - s) * g2; sign = -1.0; add = false;      // S(i+s, i-s) for the next level (stride 2 s)
  } else {
    if (!has_up) return;
    v = bcr_gemm_tile<false, false>(Kup + (size_t)i * g2, Yup + (size_t)i * g2, gs, m0, n0, li, lk);
    dst = D + (size_t)(i + s) * g2; sign = -1.0; add = true;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = m0 + lk + 4 * r, col = n0 + li;
    if (row < gs && col < gs) {
      const size_t idx = (size_t)row + (size_t)col * gs;
      dst[idx] = add ? dst[idx] + sign * v[r] : sign * v[r];
    }
  }
}

__global__ __launch_bounds__(256) void k_btd_finish(int G, const int* __restrict__ infos, int* __restrict__ bkinfo,
                                                    const double* __restrict__ tail, const int* __restrict__ scatter_err,
                                                    long long* status_out, long long seq) {
  __shared__ int part[3][4];
  if (blockIdx.x != 0) return;
  int pos = 0, neg = 0, zero = 0;
  for (int t = threadIdx.x; t < G; t += 256) { pos += infos[4 * t]; neg += infos[4 * t + 1]; zero += infos[4 * t + 2]; }
  for (int off = 32; off > 0; off >>= 1) { pos += __shfl_xor(pos, off); neg += __shfl_xor(neg, off); zero += __shfl_xor(zero, off); }
  if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = pos; part[1][threadIdx.x >> 6] = neg; part[2][threadIdx.x >> 6] = zero; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  for (int k = 0; k < 3; ++k) bkinfo[k] = part[k][0] + part[k][1] + part[k][2] + part[k][3];
  publish_status(tail, bkinfo, status_out, seq);
  if (scatter_err[0]) { status_out[0] = 3; }
}

// y[r] = sum_k M[r + k gs] v[k] (r < gs <= 512) by all threads of a 512-thread workgroup: the K range is dealt over the
// blockDim / RP groups of RP = gs rounded up to 64 threads, the partial sums meet in LDS and are added in group order
// (deterministic).  With thread = row and a loop over all of K only gs of the 512 threads worked, each through gs dependent
// additions (35 us per launch at gs = 98).  Returns the sum in the threads of group 0 (tid < gs); all threads must call.
__device__ __forceinline__ double bcr_matvec(const double* __restrict__ M, const double* __restrict__ v, int gs, double (*part)[512]) {
  const int RP = (gs + 63) / 64 * 64, np = max(1, (int)blockDim.x / RP);
  const int r = (int)threadIdx.x % RP, grp = (int)threadIdx.x / RP;
  double a0 = 0.0, a1 = 0.0;
  if (r < gs && grp < np) {
    int k = grp;
    for (; k + np < gs; k += 2 * np) { a0 += M[(size_t)r + (size_t)k * gs] * v[k]; a1 += M[(size_t)r + (size_t)(k + np) * gs] * v[k + np]; }
    if (k < gs) a0 += M[(size_t)r + (size_t)k * gs] * v[k];
  }
  if (grp < np && grp < 8) part[grp][r] = a0 + a1;
  __syncthreads();
  double y = 0.0;
  if (grp == 0 && r < gs)
    for (int q = 0; q < min(np, 8); ++q) y += part[q][r];
  __syncthreads();
  return y;
}
// y[r] = sum_k M[k + r gs] v[k] (the transposed block): a wave per row, lanes along k (contiguous), shuffle reduction; the
// rows r = wave, wave + nwaves, ...  Result of row r in out[r] (LDS), complete after the barrier.
__device__ __forceinline__ void bcr_matvec_t(const double* __restrict__ M, const double* __restrict__ v, int gs, double* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int r = wave; r < gs; r += nw) {
    double a = 0.0;
    for (int k = lane; k < gs; k += 64) a += M[(size_t)k + (size_t)r * gs] * v[k];
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
    if (lane == 0) out[r] = a;
  }
  __syncthreads();
}

// solve, forward part of a level: phase 0: u_i = inv_i b_i (kept in w); phase 1: b_{i-s} -= Klo_i^T u_i;
// phase 2: b_{i+s} -= Kup_i u_i.  One workgroup per eliminated block.
__global__ __launch_bounds__(512) void k_bcr_fwd(int gs, int G, BcrLevel lv, int phase, const double* __restrict__ inv,
                                                 const double* __restrict__ Klo, const double* __restrict__ Kup,
                                                 double* __restrict__ b, double* __restrict__ w) {
  __shared__ double part[8][512];
  const int i = lv.elim[blockIdx.x], s = lv.s, r = threadIdx.x;
  const size_t g2 = (size_t)gs * gs;
  if (phase == 0) {
    const double a = bcr_matvec(inv + (size_t)i * g2, b + (size_t)i * gs, gs, part);
    if (r < gs) w[(size_t)i * gs + r] = a;
  } else if (phase == 1) {
    if (!lv.lo || i - s < 0) return;
    bcr_matvec_t(Klo + (size_t)i * g2, w + (size_t)i * gs, gs, part[0]);
    if (r < gs) b[(size_t)(i - s) * gs + r] -= part[0][r];
  } else {
    if (i + s >= G) return;
    const double a = bcr_matvec(Kup + (size_t)i * g2, w + (size_t)i * gs, gs, part);
    if (r < gs) b[(size_t)(i + s) * gs + r] -= a;
  }
}

// solve, backward part of a level: x_i = u_i - Ylo_i x_{i-s} - Yup_i x_{i+s}
__global__ __launch_bounds__(512) void k_bcr_bwd(int gs, int G, BcrLevel lv, const double* __restrict__ Ylo,
                                                 const double* __restrict__ Yup, const double* __restrict__ w,
                                                 double* __restrict__ x) {
  __shared__ double part[8][512];
  const int i = lv.elim[blockIdx.x], s = lv.s, r = threadIdx.x;
  const size_t g2 = (size_t)gs * gs;
  double a = (r < gs) ? w[(size_t)i * gs + r] : 0.0;
  if (lv.lo && i - s >= 0) a -= bcr_matvec(Ylo + (size_t)i * g2, x + (size_t)(i - s) * gs, gs, part);
  if (i + s < G) a -= bcr_matvec(Yup + (size_t)i * g2, x + (size_t)(i + s) * gs, gs, part);
  if (r < gs) x[(size_t)i * gs + r] = a;
}

__global__ __launch_bounds__(256) void k_bcr_rhs(int n, const double* rc, const double* rs, double* b) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) b[i] = (rc ? rc[i] : 0.0) + rs[i];
}

// xc = S^-1 (rc + rs): blocked LDL^T factor if it was accepted, else the Bunch-Kaufman factor
// THREADS x NBS: 512 threads with 32-column segments (n_c <= 512), or 1024 threads with 16-column segments
// (512 < n_c <= 1024: two 16-entry segments are what 128 VGPRs per thread leave room for; the global-memory
// fallback ldl_blocked_solve puts three dependent load round trips into each of its 2 n_c / 32 block steps:
// 0.89 ms at n_c = 1000)
template <int THREADS, int NBS>
__global__ __launch_bounds__(THREADS) void k_coupling_solve(int n, const double* Abk, const int* ipiv,
                                                            const double* Aldl, const double* dvec, const int* mode,
                                                            const double* rc, const double* rs, double* xc) {
  extern __shared__ __attribute__((aligned(16))) double xs[];
  __shared__ double sv[16];
  __shared__ int si[16];
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

// ------------------------------------------------------------------------------------------
// gather of one scalar row: sum over entries of U[upos] * Z[zcol]; the (upos, zcol) records are
// fetched with one vector load per 64 entries and broadcast, loads issue in groups of 16
// Z operand of a solve entry: row zc of Y, or -- native right-hand sides, zc < 0 -- row -1 - zc of the caller's
// right-hand side (a column without incoming entries is never written to Y: y = b there)
#define PP_ZVAL(zc) (((zc) >= 0) ? Z[(size_t)(zc) * bpad] : RN[(size_t)(-1 - (zc)) * bpad])

__device__ __forceinline__ double gather_row(const int* __restrict__ upos, const int* __restrict__ zcol, int e0, int e1,
                                             const double* __restrict__ U, const double* __restrict__ Z,
                                             const double* __restrict__ RN, size_t bpad, int lane) {
  double s0 = 0.0, s1 = 0.0;
  if (e1 - e0 <= 3) {   // wide bottom levels: one to three entries, plain scalar record reads
    for (int e = e0; e < e1; ++e) { const int zc = zcol[e]; s0 += U[(size_t)upos[e] * bpad] * PP_ZVAL(zc); }
    return s0;
  }
  for (int eb = e0; eb < e1; eb += 64) {
    const int cnt = min(64, e1 - eb);
    int ru = 0, rz = 0;
    if (lane < cnt) { ru = upos[eb + lane]; rz = zcol[eb + lane]; }
#define PP_GGROUP(G)                                                                       \
  {                                                                                        \
    int iu[G], iz[G];                                                                      \
    _Pragma("unroll") for (int i = 0; i < G; ++i) {                                        \
      const int q = min(i0 + i, cnt - 1);                                                  \
      iu[i] = bcast(ru, q); iz[i] = bcast(rz, q);                                          \
    }                                                                                      \
    double u[G], z[G];                                                                     \
    _Pragma("unroll") for (int i = 0; i < G; ++i) { u[i] = U[(size_t)iu[i] * bpad]; z[i] = PP_ZVAL(iz[i]); } \
    _Pragma("unroll") for (int i = 0; i < G; i += 2) {                                     \
      s0 += (i0 + i < cnt) ? u[i] * z[i] : 0.0;                                            \
      s1 += (i0 + i + 1 < cnt) ? u[i + 1] * z[i + 1] : 0.0;                                \
    }                                                                                      \
  }
    int i0 = 0;
    for (; cnt - i0 > 4; i0 += 16) PP_GGROUP(16)
    if (i0 < cnt) PP_GGROUP(4)
#undef PP_GGROUP
  }
  return s0 + s1;
}

// Wave team: the NW waves of a workgroup each hold a partial sum of ONE row (a slice of its entry list); the sums
// meet in LDS and are added in wave order (deterministic).  Returns the total in wave 0 (others: unspecified).
// The long rows / columns of the top levels -- up to a few hundred entries, one dependent load round per 16 of
// them -- set the duration of their launches; a team cuts the rounds by NW.
template <int NW>
__device__ __forceinline__ double team_sum(double s, double (*red)[64], int wave, int lane) {
  if (NW == 1) return s;
  red[wave][lane] = s;
  __syncthreads();
  double t = 0.0;
  if (wave == 0) {
#pragma unroll
    for (int k = 0; k < NW; ++k) t += red[k][lane];
  }
  return t;
}

// forward substitution, one scalar row per workgroup: y_c = b_c - sum_k L[c, k] y_k
// (rows of one block pivot are independent: the block is applied as a whole, by inv(P), in the backward sweep)
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_fwd_level(GroupDev g, int col0, int chunk0, int ny) {
  __shared__ double red[NW][64];
  const int lane = threadIdx.x & 63, wave = (NW > 1) ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
  const int b = (PP_CHUNK_OF_WG(ny) + chunk0) * 64 + lane;
  const size_t bpad = (size_t)g.bpad;
  const int* rec = g.fwd_rec + 4 * (size_t)(col0 + PP_TASK_OF_WG(ny));   // {column, -, e0, e1}
  const int c = rec[0], e0 = rec[2], ne = rec[3] - rec[2];
  const int a0 = e0 + (int)((long long)ne * wave / NW), a1 = e0 + (int)((long long)ne * (wave + 1) / NW);
  // in place: the right-hand side was transposed into Y in the new order (rows of leaf columns are final as they are);
  // native right-hand sides: y_c = b_(original row of c) - s, read where the caller left it
  double* yc = g.Y + (size_t)c * bpad + b;
  const double* RN = g.rhsN ? g.rhsN + b : nullptr;
  const double s = team_sum<NW>(gather_row(g.sfwd_upos, g.sfwd_zcol, a0, a1, g.L + b, g.Y + b, RN, bpad, lane), red, wave, lane);
  if (wave == 0) *yc = (RN ? RN[(size_t)rec[1] * bpad] : *yc) - s;
}

// The wide bottom levels with two instances per lane (chunks of 128 instances, 16-byte accesses): short rows only
// (the callers use it on levels whose longest row or column has at most PP_PAIR_MAXROW entries; one wave per row, no team).
constexpr int PP_PAIR_MAXROW = 16;   // (8: backward sweep 0.194 ms, 16: 0.169, 32: 0.177 at C3)
__global__ __launch_bounds__(64) void k_fwd_level_pair(GroupDev g, int col0, int chunk0, int ny) {
  const int lane = threadIdx.x;
  const unsigned b = (unsigned)(((PP_PAIR_OF_WG(ny) + chunk0) * 64 + lane) * 2);
  const size_t bpad = (size_t)g.bpad;
  const int* rec = g.fwd_rec + 4 * (size_t)(col0 + PP_TASK_OF_WG(ny));   // {column, original row, e0, e1}
  const int c = rec[0], e0 = rec[2], e1 = rec[3];
  const double* __restrict__ Lb = g.L + b;
  const double* __restrict__ Z = g.Y + b;
  const double* __restrict__ RN = g.rhsN ? g.rhsN + b : nullptr;
  double s[2] = {0.0, 0.0};
  for (int eb = e0; eb < e1; eb += 4) {
    double u[4][2], z[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = min(eb + i, e1 - 1);
      const int zc = g.sfwd_zcol[e];
      ldv<2>(Lb + (size_t)g.sfwd_upos[e] * bpad, u[i]);
      ldv<2>((zc >= 0) ? Z + (size_t)zc * bpad : RN + (size_t)(-1 - zc) * bpad, z[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (eb + i < e1) { s[0] += u[i][0] * z[i][0]; s[1] += u[i][1] * z[i][1]; }
    }
  }
  double* yc = g.Y + (size_t)c * bpad + b;
  double y0[2];
  ldv<2>(RN ? RN + (size_t)rec[1] * bpad : yc, y0);
  const double out[2] = {y0[0] - s[0], y0[1] - s[1]};
  stv<2>(yc, out);
}

__global__ __launch_bounds__(64) void k_bwd_level_pair(GroupDev g, int col0, int chunk0, int ny, const double* __restrict__ xc) {
  const int lane = threadIdx.x;
  const unsigned b = (unsigned)(((PP_PAIR_OF_WG(ny) + chunk0) * 64 + lane) * 2);
  const size_t bpad = (size_t)g.bpad;
  const int* rec = g.bwd_rec + 8 * (size_t)(col0 + PP_TASK_OF_WG(ny));   // {c, w, q, nr, rowptr, L base, doff, p0}
  const int c = rec[0], w = rec[1], q = rec[2], nr = rec[3];
  const int* ri = g.rowidx + rec[4];
  const double* Lp = g.L + (size_t)rec[5] * bpad + b;   // column q of the rows below the block
  const double* Xb = g.X + b;
  const int n = g.n;
  const size_t rstride = (size_t)w * bpad;
  double z[2] = {0.0, 0.0};
  {
    const double* inv = g.Dinv + (size_t)rec[6] * bpad + b;
    const int p0 = rec[7] >= 0 ? rec[7] : -1 - rec[7];
    const double* Yp = g.Y + (size_t)p0 * bpad + b;
#pragma unroll
    for (int t = 0; t < PP_WMAX; ++t) {
      if (t < w) {
        const int hi = q > t ? q : t, lo = q > t ? t : q;
        double yv[2], iv[2];
        ldv<2>(rec[7] >= 0 ? Yp + (size_t)t * bpad : g.rhsN + (size_t)g.perm[p0 + t] * bpad + b, yv);
        ldv<2>(inv + (size_t)(hi * (hi + 1) / 2 + lo) * bpad, iv);
        z[0] += iv[0] * yv[0]; z[1] += iv[1] * yv[1];
      }
    }
    for (int t = PP_WMAX; t < w; ++t) {        // (only the root front is wider than PP_WMAX)
      const int hi = q > t ? q : t, lo = q > t ? t : q;
      double yv[2], iv[2];
      ldv<2>(rec[7] >= 0 ? Yp + (size_t)t * bpad : g.rhsN + (size_t)g.perm[p0 + t] * bpad + b, yv);
      ldv<2>(inv + (size_t)(hi * (hi + 1) / 2 + lo) * bpad, iv);
      z[0] += iv[0] * yv[0]; z[1] += iv[1] * yv[1];
    }
  }
  double s[2] = {0.0, 0.0};
  for (int jb = 0; jb < nr; jb += 4) {
    double u[4][2], x[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = min(jb + i, nr - 1);
      const int r = ri[j];
      ldv<2>(Lp + (size_t)j * rstride, u[i]);
      if (r < n) ldv<2>(Xb + (size_t)r * bpad, x[i]);
      else {
        x[i][0] = xc[(size_t)(r - n) * g.xs_row + (size_t)b * g.xs_lane];
        x[i][1] = xc[(size_t)(r - n) * g.xs_row + (size_t)(b + 1) * g.xs_lane];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (jb + i < nr) { s[0] += u[i][0] * x[i][0]; s[1] += u[i][1] * x[i][1]; }
    }
  }
  const double out[2] = {((int)b < g.batch) ? z[0] - s[0] : 0.0, ((int)b + 1 < g.batch) ? z[1] - s[1] : 0.0};
  stv<2>(g.X + (size_t)c * bpad + b, out);
}

// coupling row c: rspart[chunk][c] = - sum over active instances and panels of L[c,k] y_k
// (one wave per row: 200 rows x 16 chunks fill the chip, and a team of waves per row measured slower here)
__global__ __launch_bounds__(64) void k_fwd_coupling(GroupDev g, double* __restrict__ rs_mapped) {
  const int lane = threadIdx.x;
  const int chunk = PP_CHUNK_OF_WG(g.nchunk);
  const int b = chunk * 64 + lane;
  const size_t bpad = (size_t)g.bpad;
  const int c = PP_TASK_OF_WG(g.nchunk);
  double s = -gather_row(g.crow_upos, g.crow_zcol, g.crow_eptr[c], g.crow_eptr[c + 1], g.L + b, g.Y + b,
                         g.rhsN ? g.rhsN + b : nullptr, bpad, lane);
  if (b >= g.batch) s = 0.0;
  if (g.cmapT) {      // mapped group: every instance adds to coupling rows of its own
    if (b < g.batch && s != 0.0) atomicAdd(&rs_mapped[g.cmapT[(size_t)c * bpad + b]], s);
    return;
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) g.rspart[(size_t)chunk * g.nc + c] = s;
}

__global__ __launch_bounds__(256) void k_rs_reduce(GroupDev g, double* __restrict__ rs, int store) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= g.nc) return;
  double s = 0.0;
  for (int q = 0; q < g.nchunk; ++q) s += g.rspart[(size_t)q * g.nc + c];
  rs[c] = store ? s : rs[c] + s;     // (the first group stores: no memset of r_s in front of the sweep)
}

// back substitution, one scalar column c = (block pivot p, component q) per workgroup:
//   x_c = (inv(P_p) y_p)_q - sum_i L[i, c] x_i      (x_i = xc for coupling rows)
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_bwd_level(GroupDev g, int col0, int chunk0, int ny,
                                                       const double* __restrict__ xc) {
  __shared__ double red[NW][64];
  const int lane = threadIdx.x & 63, wave = (NW > 1) ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
  const int b = (PP_CHUNK_OF_WG(ny) + chunk0) * 64 + lane;
  const size_t bpad = (size_t)g.bpad;
  const int* rec = g.bwd_rec + 8 * (size_t)(col0 + PP_TASK_OF_WG(ny));   // {c, w, q, nr, rowptr, L base, doff, p0}
  const int c = rec[0], w = rec[1], q = rec[2], nr = rec[3];
  const int* ri = g.rowidx + rec[4];
  const double* Lp = g.L + (size_t)rec[5] * bpad + b;   // column q of the rows below the block
  const double* Xb = g.X + b;
  const int n = g.n;
  const size_t rstride = (size_t)w * bpad;
  // z = row q of inv(P) times y_p
  double z = 0.0;
  if (wave == 0) {
    const double* inv = g.Dinv + (size_t)rec[6] * bpad + b;
    // rec[7] = first column p0 of the block, or -1 - p0 if its columns have no incoming entries (native right-hand
    // sides only: y = b there, read through the permutation from the caller's rows)
    const int p0 = rec[7] >= 0 ? rec[7] : -1 - rec[7];
    const double* Yp = g.Y + (size_t)p0 * bpad + b;
#pragma unroll
    for (int t = 0; t < PP_WMAX; ++t) {
      if (t < w) {
        const int hi = q > t ? q : t, lo = q > t ? t : q;
        const double yv = rec[7] >= 0 ? Yp[(size_t)t * bpad] : g.rhsN[(size_t)g.perm[p0 + t] * bpad + b];
        z += inv[(size_t)(hi * (hi + 1) / 2 + lo) * bpad] * yv;
      }
    }
    if (w > PP_WMAX) {                         // (only the root front: all its loads in flight together)
      double yv[PP_FRONT_MAX], iv[PP_FRONT_MAX];
#pragma unroll
      for (int t = PP_WMAX; t < PP_FRONT_MAX; ++t) {
        const int tt = min(t, w - 1), hi = q > tt ? q : tt, lo = q > tt ? tt : q;
        yv[t] = rec[7] >= 0 ? Yp[(size_t)tt * bpad] : g.rhsN[(size_t)g.perm[p0 + tt] * bpad + b];
        iv[t] = inv[(size_t)(hi * (hi + 1) / 2 + lo) * bpad];
      }
#pragma unroll
      for (int t = PP_WMAX; t < PP_FRONT_MAX; ++t) z += (t < w) ? iv[t] * yv[t] : 0.0;
    }
  }
  const int j0 = (int)((long long)nr * wave / NW), j1 = (int)((long long)nr * (wave + 1) / NW);   // this wave's rows
  double g0 = 0.0, g1 = 0.0;
  if (NW == 1 && nr <= 4) {   // wide bottom levels: short panels, plain scalar index reads
    for (int j = 0; j < nr; ++j) {
      const int r = ri[j];
      g0 += Lp[(size_t)j * rstride] * ((r < n) ? Xb[(size_t)r * bpad] : xc[(size_t)(r - n) * g.xs_row + (size_t)b * g.xs_lane]);
    }
  } else {
    for (int jb = j0; jb < j1; jb += 64) {
      const int cnt = min(64, j1 - jb);
      const int rv = (lane < cnt) ? ri[jb + lane] : 0;
#define PP_BGROUP(G)                                                                       \
  {                                                                                        \
    int rr[G];                                                                             \
    _Pragma("unroll") for (int i = 0; i < G; ++i) rr[i] = bcast(rv, min(i0 + i, cnt - 1)); \
    double u[G], x[G];                                                                     \
    _Pragma("unroll") for (int i = 0; i < G; ++i) {                                        \
      const int qq = min(i0 + i, cnt - 1);                                                 \
      u[i] = Lp[(size_t)(jb + qq) * rstride];                                              \
      x[i] = (rr[i] < n) ? Xb[(size_t)rr[i] * bpad] : xc[(size_t)(rr[i] - n) * g.xs_row + (size_t)b * g.xs_lane]; \
    }                                                                                      \
    _Pragma("unroll") for (int i = 0; i < G; i += 2) {                                     \
      g0 += (i0 + i < cnt) ? u[i] * x[i] : 0.0;                                            \
      g1 += (i0 + i + 1 < cnt) ? u[i + 1] * x[i + 1] : 0.0;                                \
    }                                                                                      \
  }
      int i0 = 0;
      for (; cnt - i0 > 4; i0 += 16) PP_BGROUP(16)
      if (i0 < cnt) PP_BGROUP(4)
#undef PP_BGROUP
    }
  }
  const double s = team_sum<NW>(g0 + g1, red, wave, lane);
  if (wave == 0) g.X[(size_t)c * bpad + b] = (b < g.batch) ? z - s : 0.0;    // (padded lanes of a ragged chunk stay zero)
}


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

// ------------------------------------------------------------------------------------------
struct Group {
  pp::Plan plan;
  GroupDev dev;
  int batch = 0, nraw = 0;
  std::vector<int> can_ptr, can_idx;
  std::vector<void*> allocs;
  int ntiles = 0;
  int nmt = 0, nmt_items = 0;  // 16 x 16 tiles of S with contributions (MFMA form), work items over them
  bool mt_wide = false;        // 32 x 32 super-tiles (k_schur_mfma_wide): nmt counts quarters, a work item holds four partial tiles
  double *raw_own = nullptr, *rhs_own = nullptr, *rawT_own = nullptr;
  int nraw_used = 0;
  std::vector<int> level_maxw;   // widest block pivot per level (selects the scalar kernel variants)
  const int* wtask = nullptr;    // scale chunks of the root front (device), plan.wtasks
  double* front_inv = nullptr;   // root front: inv(P) as a zero-padded 16 x 16 matrix [entry][instance] (k_front_invert -> k_scale_wide)
  std::vector<int> diag_can;     // canonical entry of the diagonal (i, i) of K, or -1
  std::vector<uint8_t> fwd_level_has_entries;   // forward-solve levels whose columns have any incoming entry
  std::vector<int> fwd_level_team, bwd_level_team;   // waves per row / column on each solve level (1, 4 or 16)
  std::vector<int> fwd_level_maxrow, bwd_level_maxrow;   // longest row / column of the level (entries)
  int nraw_tiles = 0;            // number of input tiles with needed entries
  int nshift = 0;                // rows with a regularisation class (pp_set_diagonal_classes)
  int *shift_row = nullptr, *shift_cls = nullptr;   // device: transposed-input row of their diagonal entry, class
  std::vector<void*> value_allocs;   // value storage (raw, rawT, U, L, ...): allocated by alloc_value_storage
  // where the next numeric factorisation takes its values from
  enum { IN_RAW = 0, IN_COMPACT = 1, IN_SOURCES = 2 };
  int input_mode = IN_RAW;
  std::vector<int> used_raw;         // raw entries some canonical entry reads, ascending: row j of rawT is raw entry used_raw[j]
  int nsrc = 0;                      // f2 value map: rows of the source buffer
  int *map_src = nullptr;            // device [nraw_used]: source row of each used raw entry, or -1 (constant)
  double *map_coef = nullptr;        // device [nraw_used]
  double *src_own = nullptr, *src = nullptr;   // [nsrc][bpad]
  std::vector<int> fent_host, init_rec;        // entry records as uploaded; positions of the initial-value records
  int *fent_src = nullptr;                     // device: the same records with the initial-value ones pointing at sources
  double *xout_own = nullptr;
  // native [row][instance] vectors (pp_bind_native_vectors): the sweeps read b / write x where the caller keeps them
  const int *zcolN_f = nullptr, *zcolN_c = nullptr, *brecN = nullptr, *rowidx_o = nullptr;
  const double* rhs_native = nullptr;
  double* x_native = nullptr;
  int nc_loc = 0;                    // coupling rows of the group's plan (== n_c unless the group is mapped)
  std::vector<int> cmap_host;        // mapped group: [batch][nc_loc] global coupling indices
};

}  // namespace

// Host threads that enqueue the launches of pattern groups side by side (one per auxiliary group stream; the caller's
// thread takes stream 0).  A time-staged problem issues ~1000 small launches per step over three streams: with one
// enqueuing thread the step is bound by the host's launch rate on a slow host (15.3 ms through the interface against
// 10.5 ms of kernels at C4), not by the GPU.  Created at first use, parked on a condition variable in between.
struct EnqueuePool {
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv, cv_done;
  std::function<void(int)> job;
  long long gen = 0;
  int nwork = 0, pending = 0;
  bool stop = false;
  void worker(int k, int device) {
    (void)hipSetDevice(device);
    long long seen = 0;
    for (;;) {
      std::function<void(int)> f;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return stop || (gen != seen && k <= nwork); });
        if (stop) return;
        seen = gen;
        f = job;
      }
      f(k);
      {
        std::lock_guard<std::mutex> lk(m);
        if (--pending == 0) cv_done.notify_all();
      }
    }
  }
  // runs f(0) on the caller and f(1) .. f(n - 1) on workers; returns when all are done
  void run(int n, int device, const std::function<void(int)>& f) {
    while ((int)th.size() < n - 1) { const int k = (int)th.size() + 1; th.emplace_back([this, k, device] { worker(k, device); }); }
    {
      std::lock_guard<std::mutex> lk(m);
      job = f; nwork = n - 1; pending = n - 1; ++gen;
    }
    cv.notify_all();
    f(0);
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [&] { return pending == 0; });
  }
  ~EnqueuePool() {
    { std::lock_guard<std::mutex> lk(m); stop = true; }
    cv.notify_all();
    for (std::thread& t : th) t.join();
  }
};

struct pp_solver {
  int device = 0;
  hipStream_t stream = nullptr;
  int nc = 0;
  bool symbolic_done = false, blocks_factored = false, numeric_done = false, schur_done = false;
  std::vector<Group*> groups;
  double *S = nullptr, *S_own = nullptr, *Sfac = nullptr, *Sldl = nullptr, *dvec = nullptr, *Qd = nullptr, *work = nullptr;
  int* dense_mode = nullptr;
  int dense_policy = 0;   // 0 auto (optimistic blocked LDL^T, Bunch-Kaufman fallback), 1 Bunch-Kaufman only
  double *rs = nullptr, *rs_own = nullptr, *rcd = nullptr, *xc = nullptr;
  int *ipiv = nullptr, *bkinfo = nullptr, *counters = nullptr;
  // status mailbox in pinned, device-mapped host memory: {status, pos, neg, zero, sequence}; the last
  // kernel of pp_factor_schur writes it, pp_get_status polls the sequence word (no stream sync, no copies)
  volatile long long* status_host = nullptr;
  long long* status_dev = nullptr;
  long long status_seq = 0;
  double fail_code = 0.0;
  double* vec_part = nullptr;    // scratch of the f4 vector kernels
  // coupling structure: dense S (default) or block-tridiagonal with G blocks of gs rows (n_c = G * gs)
  int btd = 0, gs = 0, G = 0;
  double *btd_fac = nullptr, *btd_inv = nullptr, *btd_x = nullptr, *btd_q = nullptr, *btd_vec = nullptr;
  double *btd_klo = nullptr, *btd_kup = nullptr, *btd_ylo = nullptr, *btd_yup = nullptr;
  int *btd_ipiv = nullptr, *btd_info = nullptr, *scatter_err = nullptr, *btd_elim = nullptr;
  std::vector<int> bcr_off, bcr_ne, bcr_s, bcr_lo;   // per level: offset into btd_elim, eliminated blocks, stride, lower neighbour live
  int btd_sequential = 0;
  bool bcr_lds_attr = false, bcr_ldl_attr = false;
  // largest multiplier the unpivoted block factorisation of the cyclic reduction accepts (1 / u, u = 0.01; PP_BCR_LBOUND:
  // test switch -- a bound below 1 sends some blocks to Bunch-Kaufman and leaves others on the unpivoted path)
  double bcr_lbound = std::getenv("PP_BCR_LBOUND") ? std::atof(std::getenv("PP_BCR_LBOUND")) : 100.0;
  double growth_bound = 1e8;     // 1 / u_rt: a factor entry beyond it flags its instance
  bool growth_fatal = false;     // flagged instances make the factorisation report status 2 (else they are only counted)
  double pivot_threshold = 0.0;  // symbolic-time threshold u for groups added afterwards (0: plan default)
  bool no_fused_sources = std::getenv("PP_NO_FUSED_SOURCES") != nullptr;   // measurement switch: assemble the sources first
  bool schur_mfma = std::getenv("PP_NO_SCHUR_MFMA") == nullptr;   // MFMA form of the Schur update of unmapped groups (measurement switch)
  bool enqueue_threads = std::getenv("PP_NO_ENQUEUE_THREADS") == nullptr;   // one enqueuing host thread per group stream (measurement switch)
  EnqueuePool pool;
  long long* corner_pos = nullptr;      // sparse Q of a block-tridiagonal S: positions in the Schur layout, values
  double* corner_val = nullptr;
  size_t corner_cap = 0;
  hipEvent_t ev_corner_up = nullptr, ev_corner_done = nullptr;
  hipStream_t up_stream = nullptr;
  bool corner_used = false;
  std::mutex alloc_mu, err_mu;
  bool group_streams = std::getenv("PP_NO_GROUP_STREAMS") == nullptr;   // pattern groups side by side on streams of their own (measurement switch)
  bool dense_dpp = std::getenv("PP_NO_DENSE_DPP") == nullptr;           // row broadcasts by DP-ALU DPP in k_ldl_regs (measurement switch)
  bool lane_pairs = std::getenv("PP_NO_LANE_PAIRS") == nullptr;   // two instances per lane in the gather kernels (measurement switch)
  double shift_w = 0.0, shift_c = 0.0;   // diagonal shifts of the current pp_numeric_local_shifted call (else 0)
  double mem_factor = 1.0;
  int64_t mem_budget = 0;        // bytes of device value storage the handle may allocate (0: no limit); scaled by mem_factor
  int64_t mem_required = 0;      // bytes of value storage the current plan needs
  bool values_allocated = false;
  std::string err;
  // Instance groups ("splits"): the level sweeps of disjoint 64-instance chunk ranges are
  // independent and can be issued on separate streams.  Measured (C3, 1 GPU, 4 splits): the 4x
  // launches serialise instead of overlapping (254 vs 379 it/s), so the default is one split.
  int nsplit_req = 0;   // 0 = default (1)
  int sn_wmax = 0, sn_tol = -1;   // supernode options for groups added afterwards (0 / -1: plan defaults)
  hipStream_t aux[PP_MAX_SPLIT] = {};
  hipEvent_t ev_fork = nullptr, ev_join[PP_MAX_SPLIT] = {};
  bool aux_made = false;
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

int build_btd_schedule(pp_handle h);

size_t schur_doubles(pp_handle h) {
  return h->btd ? (size_t)(2 * h->G - 1) * h->gs * h->gs : (size_t)h->nc * h->nc;
}

int fail(pp_handle h, int status, const std::string& msg) {
  if (h) { std::lock_guard<std::mutex> lk(h->err_mu); h->err = msg; }
  return status;
}

// Elimination schedule of the block-tridiagonal S.  Cyclic reduction (default): level l eliminates the blocks
// i = s (2k + 1), s = 2^l, block 0 goes last.  Sequential (btd_sequential): one block per level in ascending order, each
// coupled only to its upper neighbour -- the fallback when a diagonal block of the odd-even order is singular (an
// indefinite S has singular principal submatrices; the ascending order is the forward sweep of the time-staged problem).
int build_btd_schedule(pp_handle h) {
  std::vector<int> elim;
  h->bcr_off.clear(); h->bcr_ne.clear(); h->bcr_s.clear(); h->bcr_lo.clear();
  if (h->btd_sequential) {
    for (int t = 0; t < h->G; ++t) {
      h->bcr_off.push_back(t); h->bcr_ne.push_back(1); h->bcr_s.push_back(1); h->bcr_lo.push_back(0);
      elim.push_back(t);
    }
  } else {
    int sdt = 1;
    for (; sdt < h->G; sdt *= 2) {
      h->bcr_off.push_back((int)elim.size());
      int ne = 0;
      for (int i = sdt; i < h->G; i += 2 * sdt) { elim.push_back(i); ++ne; }
      h->bcr_ne.push_back(ne);
      h->bcr_s.push_back(sdt);
      h->bcr_lo.push_back(1);
    }
    h->bcr_off.push_back((int)elim.size());
    elim.push_back(0);
    h->bcr_ne.push_back(1);
    h->bcr_s.push_back(sdt);
    h->bcr_lo.push_back(1);
  }
  if (h->btd_elim) { (void)hipFree(h->btd_elim); h->btd_elim = nullptr; }
  void* p = nullptr;
  if (hipMalloc(&p, std::max<size_t>(elim.size(), 1) * sizeof(int)) != hipSuccess) return fail(h, 1, "hipMalloc failed (schedule)");
  h->btd_elim = (int*)p;
  if (hipMemcpy(h->btd_elim, elim.data(), elim.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
    return fail(h, 3, "hipMemcpy failed (schedule)");
  return 0;
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

// Chunk ranges of the splits of a group with `nchunk` 64-instance chunks.
struct Splits {
  int n = 1;
  int c0[PP_MAX_SPLIT + 1] = {0};
};

Splits make_splits(pp_handle h, int nchunk) {
  Splits sp;
  int want = h->nsplit_req > 0 ? h->nsplit_req : 1;   // measured on MI355X/ROCm 7: splits > 1 serialise, default off
  want = std::max(1, std::min(std::min(want, PP_MAX_SPLIT), nchunk));
  sp.n = want;
  for (int i = 0; i <= want; ++i) sp.c0[i] = (int)((int64_t)nchunk * i / want);
  return sp;
}

// fork the handle's stream into sp.n streams (stream 0 of the fan is the handle's own stream)
int make_aux_streams(pp_handle h) {
  if (h->aux_made) return 0;
  for (int i = 0; i < PP_MAX_SPLIT; ++i) {
    if (hipStreamCreateWithFlags(&h->aux[i], hipStreamNonBlocking) != hipSuccess) return 3;
    if (hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming) != hipSuccess) return 3;
  }
  if (hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess) return 3;
  h->aux_made = true;
  return 0;
}

// Pattern groups are independent of each other until their results meet (S, r_s): with more than one group, group gi
// runs on stream gi mod PP_MAX_SPLIT of the handle's auxiliary streams (group 0 on the handle's own), so that the
// launch chain of a small group -- a single block is 1/64 of one wave per task, but as many launches as a full group --
// hides beside the others instead of in front of them.  Not while profiling (the phase brackets sit on the handle's
// stream) and not together with instance splits.
struct GroupStreams {
  int n = 1;                       // streams in use (1: everything on the handle's stream)
  hipStream_t st[PP_MAX_SPLIT];
};

int fork_group_streams(pp_handle h, GroupStreams& gs) {
  gs.n = 1;
  gs.st[0] = h->stream;
  const int want = (int)std::min<size_t>(h->groups.size(), (size_t)PP_MAX_SPLIT);
  if (want <= 1 || h->profile || h->nsplit_req > 1 || !h->group_streams) return 0;
  if (make_aux_streams(h)) return 3;
  if (hipEventRecord(h->ev_fork, h->stream) != hipSuccess) return 3;
  for (int i = 1; i < want; ++i) {
    gs.st[i] = h->aux[i];
    if (hipStreamWaitEvent(gs.st[i], h->ev_fork, 0) != hipSuccess) return 3;
  }
  gs.n = want;
  return 0;
}

int join_group_streams(pp_handle h, const GroupStreams& gs) {
  for (int i = 1; i < gs.n; ++i) {
    if (hipEventRecord(h->ev_join[i], gs.st[i]) != hipSuccess) return 3;
    if (hipStreamWaitEvent(h->stream, h->ev_join[i], 0) != hipSuccess) return 3;
  }
  return 0;
}

// body(gi) for every group: group gi on the thread of its stream (gi mod gs.n) when the groups run side by side
template <class F>
int run_groups(pp_handle h, const GroupStreams& gs, const F& body) {
  const size_t ng = h->groups.size();
  if (gs.n <= 1 || !h->enqueue_threads) {
    for (size_t gi = 0; gi < ng; ++gi)
      if (int rc = body(gi)) return rc;
    return 0;
  }
  int rcs[PP_MAX_SPLIT] = {0};
  const int n = gs.n;
  h->pool.run(n, h->device, [&](int k) {
    for (size_t gi = (size_t)k; gi < ng; gi += (size_t)n)
      if (int rc = body(gi)) { rcs[k] = rc; break; }
  });
  for (int k = 0; k < n; ++k)
    if (rcs[k]) return rcs[k];
  return 0;
}

int fork_streams(pp_handle h, const Splits& sp, hipStream_t* out, hipStream_t base) {
  out[0] = base;
  if (sp.n == 1) return 0;
  if (make_aux_streams(h)) return 3;
  if (hipEventRecord(h->ev_fork, h->stream) != hipSuccess) return 3;
  for (int i = 1; i < sp.n; ++i) {
    out[i] = h->aux[i];
    if (hipStreamWaitEvent(out[i], h->ev_fork, 0) != hipSuccess) return 3;
  }
  return 0;
}

int join_streams(pp_handle h, const Splits& sp, hipStream_t* st) {
  for (int i = 1; i < sp.n; ++i) {
    if (hipEventRecord(h->ev_join[i], st[i]) != hipSuccess) return 3;
    if (hipStreamWaitEvent(h->stream, h->ev_join[i], 0) != hipSuccess) return 3;
  }
  return 0;
}

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

// tiles per workgroup of k_transpose_in (measured on C3: 1 is fastest -- 0.105 ms against 0.123 at 8; the
// walk along the row only pays if rows were much longer than the 4 workgroups/CU window already covers)
int transpose_tiles(int, int) {
  if (const char* e = std::getenv("PP_TRANSPOSE_TILES")) return std::max(1, std::atoi(e));
  return 1;
}

void free_group(Group* g) {
  for (void* p : {(void*)g->map_src, (void*)g->map_coef, (void*)g->src_own, (void*)g->fent_src}) if (p) (void)hipFree(p);
  if (g->shift_row) (void)hipFree(g->shift_row);
  if (g->shift_cls) (void)hipFree(g->shift_cls);
  for (void* p : g->value_allocs) (void)hipFree(p);
  for (void* p : g->allocs) (void)hipFree(p);
  delete g;
}

void free_globals(pp_handle h) {
  for (void* p : {(void*)h->S_own, (void*)h->Sfac, (void*)h->Sldl, (void*)h->dvec, (void*)h->dense_mode, (void*)h->Qd, (void*)h->work, (void*)h->rs_own, (void*)h->rcd,
                  (void*)h->xc, (void*)h->ipiv, (void*)h->bkinfo, (void*)h->counters})
    if (p) (void)hipFree(p);
  h->S = h->S_own = h->Sfac = h->Sldl = h->dvec = h->Qd = h->work = h->rs = h->rs_own = h->rcd = h->xc = nullptr;
  if (h->corner_pos) (void)hipFree(h->corner_pos);
  if (h->corner_val) (void)hipFree(h->corner_val);
  h->corner_pos = nullptr; h->corner_val = nullptr; h->corner_cap = 0; h->corner_used = false;
  h->dense_mode = nullptr;
  h->ipiv = h->bkinfo = h->counters = nullptr;
  if (h->vec_part) { (void)hipFree(h->vec_part); h->vec_part = nullptr; }
  for (void* p : {(void*)h->btd_fac, (void*)h->btd_inv, (void*)h->btd_x, (void*)h->btd_vec, (void*)h->btd_ipiv, (void*)h->btd_info,
                  (void*)h->scatter_err, (void*)h->btd_klo, (void*)h->btd_kup, (void*)h->btd_ylo, (void*)h->btd_yup, (void*)h->btd_elim})
    if (p) (void)hipFree(p);
  h->btd_fac = h->btd_inv = h->btd_x = h->btd_vec = h->btd_klo = h->btd_kup = h->btd_ylo = h->btd_yup = nullptr;
  h->btd_ipiv = h->btd_info = h->scatter_err = h->btd_elim = nullptr;
  if (h->status_host) (void)hipHostFree((void*)h->status_host);
  h->status_host = nullptr;
  h->status_dev = nullptr;
}


// bytes of value storage (everything that scales with batch x factor size) the groups of the handle need
int64_t value_storage_bytes(pp_handle h) {
  int64_t total = 0;
  for (Group* g : h->groups) {
    const pp::Plan& P = g->plan;
    const GroupDev& d = g->dev;
    const int64_t bp = d.bpad;
    // (the term magnitudes of the pivot blocks live in the rows of Y: written and read inside the factorisation, Y
    // only inside a solve)
    int64_t dbl = (int64_t)g->batch * g->nraw + (int64_t)std::max(g->nraw_used, 1) * bp + 2 * P.usize * bp +
                  (int64_t)P.dsize * bp + (int64_t)std::max(P.n + g->nc_loc, std::max(P.bsize, 1)) * bp +
                  (int64_t)P.n * bp + 2 * (int64_t)g->batch * P.n + (int64_t)d.nchunk * std::max(std::max(g->ntiles, 1) * 64, g->nmt_items * (g->mt_wide ? 1024 : 256)) +
                  (int64_t)d.nchunk * std::max(g->nc_loc, 1) +
                  (g->cmap_host.empty() ? 0 : ((int64_t)std::max(g->ntiles, 1) * 64 + std::max(g->nc_loc, 1)) * bp);
    total += 8 * dbl + 2 * (int64_t)P.npiv * bp;
  }
  return total;
}

void free_value_storage(Group* g) {
  for (void* p : g->value_allocs) (void)hipFree(p);
  g->value_allocs.clear();
  GroupDev& d = g->dev;
  d.raw = d.rawT = d.U = d.L = d.Dinv = d.Tm = d.Y = d.X = d.rhs = d.xout = d.Spart = d.rspart = nullptr;
  d.codes = nullptr;
  d.growth = nullptr;
  d.Sloc = d.XCL = nullptr;
  g->raw_own = g->rhs_own = g->xout_own = g->rawT_own = nullptr;
}

template <class T>
int value_alloc(pp_handle h, Group* g, T** out, size_t count) {
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(h, e == hipErrorOutOfMemory ? 1 : 3, std::string("hipMalloc: ") + hipGetErrorString(e));
  }
  g->value_allocs.push_back(p);
  *out = (T*)p;
  return 0;
}

// Allocates the value storage of every group if it fits the budget; status 1 (not_enough_memory) otherwise, with
// nothing left allocated (increase_memory_allocation then raises the budget and the next numeric call tries again).
int alloc_value_storage(pp_handle h) {
  if (h->values_allocated) return 0;
  if (h->mem_budget > 0 && (double)h->mem_required > (double)h->mem_budget * h->mem_factor)
    return fail(h, 1, "device value storage of " + std::to_string(h->mem_required) + " bytes exceeds the budget of " +
                          std::to_string((int64_t)((double)h->mem_budget * h->mem_factor)) +
                          " bytes (increase_memory_allocation raises it)");
  int rc = 0;
  for (Group* g : h->groups) {
    const pp::Plan& P = g->plan;
    GroupDev& d = g->dev;
    const size_t bp = (size_t)d.bpad;
    const int nc = g->nc_loc;
    // The staging copy of the input ([instance][entry]), its transposed form, and the [instance][row] copies of
    // right-hand side and solution are only needed by the input / output forms that use them (host values, host
    // vectors): ensure_optional allocates them at first use.  With device-resident sources and native vectors -- the
    // path the benchmark times -- they never exist: 0.62 of 1.68 GB at C3.  The budget is checked against the full
    // requirement here, once.
    double* keep_raw = (d.raw && d.raw != g->raw_own) ? d.raw : nullptr;   // caller-bound buffers survive
    double* keep_rhs = (d.rhs && d.rhs != g->rhs_own) ? d.rhs : nullptr;
    if ((rc = value_alloc(h, g, &d.U, (size_t)P.usize * bp))) break;
    if ((rc = value_alloc(h, g, &d.Dinv, (size_t)P.dsize * bp))) break;
    if ((rc = value_alloc(h, g, &d.L, (size_t)P.usize * bp))) break;
    // the pivot-block slots of L are never written (only the rows below the block are): define them once
    if (hipMemset(d.L, 0, (size_t)P.usize * bp * sizeof(double)) != hipSuccess) { rc = fail(h, 3, "hipMemset failed"); break; }
    if ((rc = value_alloc(h, g, &d.Y, (size_t)std::max(P.n + nc, std::max(P.bsize, 1)) * bp))) break;
    d.Tm = d.Y;     // term magnitudes of the pivot blocks (gather -> scale of one level) share the rows of the solve vector
    double* keep_x = (d.xout && d.xout != g->xout_own) ? d.xout : nullptr;
    d.xout = keep_x;
    if ((rc = value_alloc(h, g, &d.Spart, (size_t)d.nchunk * (size_t)std::max(std::max(g->ntiles, 1) * 64, g->nmt_items * (g->mt_wide ? 1024 : 256))))) break;
    if ((rc = value_alloc(h, g, &d.rspart, (size_t)d.nchunk * std::max(nc, 1)))) break;
    if (!g->cmap_host.empty()) {
      if ((rc = value_alloc(h, g, &d.Sloc, (size_t)std::max(g->ntiles, 1) * 64 * bp))) break;
      if ((rc = value_alloc(h, g, &d.XCL, (size_t)std::max(nc, 1) * bp))) break;
    }
    if ((rc = value_alloc(h, g, &d.codes, (size_t)P.npiv * bp))) break;   // 16-bit codes
    if ((rc = value_alloc(h, g, &d.growth, 2 * bp))) break;        // flags of the running factorisation | of the last one
    if (hipMemset(d.growth, 0, 2 * bp * sizeof(int)) != hipSuccess) { rc = fail(h, 3, "hipMemset failed"); break; }
    d.raw = keep_raw;
    d.rhs = keep_rhs;
  }
  if (rc) {
    const std::string msg = h->err;
    for (Group* g : h->groups) free_value_storage(g);
    h->err = msg;
    return rc;
  }
  h->values_allocated = true;
  return 0;
}

// The buffers only some input / output forms need (see alloc_value_storage); `which` is a mask.
enum { OPT_RAW = 1, OPT_RAWT = 2, OPT_RHS = 4, OPT_XOUT = 8, OPT_X = 16 };
int ensure_optional(pp_handle h, Group* g, int which) {
  std::lock_guard<std::mutex> lk(h->alloc_mu);      // (the group loops may run on several enqueuing threads)
  if (int rc = alloc_value_storage(h)) return rc;
  GroupDev& d = g->dev;
  const pp::Plan& P = g->plan;
  int rc = 0;
  if ((which & OPT_RAW) && !g->raw_own) {
    if ((rc = value_alloc(h, g, &g->raw_own, (size_t)g->batch * g->nraw))) return rc;
    if (!d.raw) d.raw = g->raw_own;
  }
  if ((which & OPT_RAWT) && !g->rawT_own) {
    if ((rc = value_alloc(h, g, &g->rawT_own, (size_t)std::max(g->nraw_used, 1) * (size_t)d.bpad))) return rc;
    d.rawT = g->rawT_own;
  }
  if ((which & OPT_RHS) && !g->rhs_own) {
    if ((rc = value_alloc(h, g, &g->rhs_own, (size_t)g->batch * P.n))) return rc;
    if (!d.rhs) d.rhs = g->rhs_own;
  }
  if ((which & OPT_X) && !d.X) {
    if ((rc = value_alloc(h, g, &d.X, (size_t)P.n * (size_t)d.bpad))) return rc;
  }
  if ((which & OPT_XOUT) && !g->xout_own) {
    if ((rc = value_alloc(h, g, &g->xout_own, (size_t)g->batch * P.n))) return rc;
    if (!d.xout) d.xout = g->xout_own;
  }
  return 0;
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
  if (h->ev_corner_up) { (void)hipEventDestroy(h->ev_corner_up); (void)hipEventDestroy(h->ev_corner_done); (void)hipStreamDestroy(h->up_stream); }
  if (h->aux_made) {
    for (int i = 0; i < PP_MAX_SPLIT; ++i) { (void)hipStreamDestroy(h->aux[i]); (void)hipEventDestroy(h->ev_join[i]); }
    (void)hipEventDestroy(h->ev_fork);
  }
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
  h->btd = 0; h->gs = h->G = 0;
  h->symbolic_done = h->blocks_factored = h->numeric_done = h->schur_done = false;
  return 0;
}

int pp_set_coupling_structure(pp_handle h, int mode, int gs, int G) {
  if (!h) return 3;
  if (h->symbolic_done || !h->groups.empty()) return fail(h, 3, "pp_set_coupling_structure: call right after pp_begin_symbolic");
  if (mode == 0) { h->btd = 0; h->gs = h->G = 0; return 0; }
  if (mode != 1 || gs < 1 || gs > 512 || G < 1 || (int64_t)gs * G != h->nc)
    return fail(h, 3, "pp_set_coupling_structure: mode 1 needs n_c = G * gs, 1 <= gs <= 512");
  h->btd = 1; h->gs = gs; h->G = G;
  return 0;
}

int pp_set_coupling_schedule(pp_handle h, int sequential) {
  if (!h) return 3;
  h->btd_sequential = sequential ? 1 : 0;
  if (h->symbolic_done && h->btd) {
    PP_HIP(hipSetDevice(h->device));
    PP_HIP(hipStreamSynchronize(h->stream));
    return build_btd_schedule(h);
  }
  return 0;
}

int64_t pp_schur_buffer_doubles(pp_handle h) { return h ? (int64_t)(schur_doubles(h) + PP_TAIL) : 0; }

int pp_add_group(pp_handle h, int n, int batch, int nnzK, const int32_t* rowK, const int32_t* colK, int nnzB,
                 const int32_t* rowB, const int32_t* colB, int nraw, const int32_t* can_ptr, const int32_t* can_idx,
                 const double* rep_vals, int* group_out) {
  if (!h) return 3;
  return pp_add_group_mapped(h, n, batch, nnzK, rowK, colK, nnzB, rowB, colB, nraw, can_ptr, can_idx, rep_vals, h->nc,
                             nullptr, group_out);
}

int pp_add_group_mapped(pp_handle h, int n, int batch, int nnzK, const int32_t* rowK, const int32_t* colK, int nnzB,
                        const int32_t* rowB, const int32_t* colB, int nraw, const int32_t* can_ptr, const int32_t* can_idx,
                        const double* rep_vals, int nc_loc, const int32_t* cmap, int* group_out) {
  if (!h) return 3;
  if (nc_loc < 0 || nc_loc > h->nc) return fail(h, 3, "pp_add_group_mapped: local coupling dimension out of range");
  if (!cmap && nc_loc != h->nc) return fail(h, 3, "pp_add_group_mapped: a group without a map uses all coupling rows");
  if (cmap)
    for (size_t i = 0; i < (size_t)batch * nc_loc; ++i)
      if (cmap[i] < 0 || cmap[i] >= h->nc) return fail(h, 3, "pp_add_group_mapped: coupling map entry out of range");
  if (h->symbolic_done) return fail(h, 3, "pp_add_group after pp_end_symbolic");
  if (batch <= 0 || n <= 0 || nraw < 0) return fail(h, 3, "bad group dimensions");
  Group* g = new Group();
  pp::PlanOptions opt;
  pp::tune_for_batch(opt, batch);
  if (h->sn_wmax > 0) opt.sn_wmax = h->sn_wmax;
  if (h->sn_tol >= 0) opt.sn_tol_rows = h->sn_tol;
  if (h->pivot_threshold > 0.0) opt.pivot_threshold = h->pivot_threshold;
  {
    std::string bad;
    if (!pp::apply_plan_tune(opt, std::getenv("PP_PLAN_TUNE"), bad)) { delete g; return fail(h, 3, "PP_PLAN_TUNE: unknown key " + bad); }
  }
  g->nc_loc = nc_loc;
  if (cmap) g->cmap_host.assign(cmap, cmap + (size_t)batch * nc_loc);
  int rc = pp::build_plan(n, nc_loc, nnzK, rowK, colK, nnzB, rowB, colB, rep_vals, opt, g->plan);
  if (rc != 0) { std::string e = g->plan.error; delete g; return fail(h, rc, "symbolic analysis failed: " + e); }
  const int ncan = nnzK + nnzB;
  g->diag_can.assign((size_t)n, -1);
  for (int e = 0; e < nnzK; ++e)
    if (rowK[e] == colK[e]) g->diag_can[(size_t)rowK[e]] = e;
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
    d.n = P.n; d.nc = g->nc_loc; d.batch = g->batch; d.bpad = (g->batch + WAVE - 1) / WAVE * WAVE;
    d.xs_row = 1; d.xs_lane = 0;
    d.rhsN = nullptr;
    d.nchunk = d.bpad / WAVE; d.npiv = P.npiv; d.nraw = g->nraw; d.usize = P.usize;
    int rc;
    std::vector<int> uoff(P.piv_uoff.begin(), P.piv_uoff.end());
    // widest column slice of a gather task per level (selects the kernel instantiation; the root front is gathered in
    // slices of PP_WMAX) and widest block pivot with ordinary scale tasks
    g->level_maxw.assign(P.n_levels, 1);
    for (int pp_ = 0; pp_ < P.npiv; ++pp_)
      g->level_maxw[P.piv_level[pp_]] = std::max(g->level_maxw[P.piv_level[pp_]], std::min(P.piv_w[pp_], PP_WMAX));
    std::vector<int> ftask, stask, fdst_ptr, fent, srec;
    const double one = 1.0;
    int one_lo, one_hi;
    { int bits[2]; std::memcpy(bits, &one, sizeof(one)); one_lo = bits[0]; one_hi = bits[1]; }
    g->init_rec.clear();
    // raw entries that some canonical entry reads get a compact row in the transposed buffer; the rest
    // (typically the upper-triangle half) are never written
    std::vector<int> rawmap((size_t)std::max(g->nraw, 1), -1);
    // compact rows follow the raw order, so that a run of needed raw entries is a run of rows: the host boundary
    // uploads only those ([batch][nraw_used], pp_upload_values_compact) and the transposition needs no row map
    for (int v : g->can_idx) rawmap[v] = 0;
    g->used_raw.clear();
    for (int e = 0; e < g->nraw; ++e) if (rawmap[(size_t)e] == 0) { rawmap[(size_t)e] = (int)g->used_raw.size(); g->used_raw.push_back(e); }
    g->nraw_used = (int)g->used_raw.size();
    // expand the canonical initial-value entries into raw-value entries (duplicates are summed)
    fdst_ptr.reserve(P.fdst_ptr.size());
    fent.reserve(P.fentries.size() * 4 + 64);
    ftask.reserve(P.ftasks.size() * TASK_INTS);
    for (auto& t : P.ftasks) {
      const int nrow = t.r1 - t.r0;
      const int new_dptr0 = (int)fdst_ptr.size();
      // fourth field of a record: bits 0-7 the destination column of an initial value or of a single-column product
      // entry (third field 0), bits 8.. the number of
      // destination rows that END before this entry (0 inside a row; k_gather_flat closes that many rows first)
      int cur_row = 0;
      for (int dd = 0; dd < nrow; ++dd) {
        fdst_ptr.push_back((int)(fent.size() / 4));
        for (int e = P.fdst_ptr[t.dptr0 + dd]; e < P.fdst_ptr[t.dptr0 + dd + 1]; ++e) {
          const pp::FEntry& fe = P.fentries[e];
          if (fe.u >= 0) { fent.insert(fent.end(), {fe.u, fe.l, fe.wk, (fe.wk == 0 ? fe.q : 0) | ((dd - cur_row) << 8)}); cur_row = dd; }
          else {
            // the L index of an initial-value record is a dummy (position 0, always valid)
            const int ce = -1 - fe.u;
            for (int q = g->can_ptr[ce]; q < g->can_ptr[ce + 1]; ++q)
            {
              g->init_rec.push_back((int)(fent.size() / 4));
              fent.insert(fent.end(), {-1 - rawmap[g->can_idx[q]], one_lo, one_hi, fe.q | ((dd - cur_row) << 8)});
              cur_row = dd;
            }
          }
        }
      }
      fdst_ptr.push_back((int)(fent.size() / 4));
      ftask.insert(ftask.end(), {t.piv, t.r0, t.r1, new_dptr0, t.kind, fdst_ptr[new_dptr0], (int)(fent.size() / 4),
                                 t.ws > 0 ? t.ws : P.piv_w[t.piv], (int)P.piv_uoff[t.piv], P.piv_boff[t.piv], P.piv_doff[t.piv],
                                 (int)P.piv_sub[t.piv], t.piece, t.npieces, P.piv_w[t.piv], t.qoff});
    }
    for (auto& t : P.stasks)
      stask.insert(stask.end(), {t.piv, t.r0, t.r1, -1, t.kind, 0, 0, P.piv_w[t.piv], (int)P.piv_uoff[t.piv],
                                 P.piv_boff[t.piv], P.piv_doff[t.piv], (int)P.piv_sub[t.piv], 0, 1, P.piv_w[t.piv], 0});
    {
      std::vector<int> wtask;
      for (auto& t : P.wtasks)
        wtask.insert(wtask.end(), {t.piv, t.r0, t.r1, -1, t.kind, 0, 0, P.piv_w[t.piv], (int)P.piv_uoff[t.piv],
                                   P.piv_boff[t.piv], P.piv_doff[t.piv], (int)P.piv_sub[t.piv], 0, 1, P.piv_w[t.piv], 0});
      g->wtask = nullptr;
      if (!wtask.empty() && (rc = dev_upload(h, g, &g->wtask, wtask))) return rc;
      g->front_inv = nullptr;
      if (P.front_piv >= 0 && (rc = dev_alloc<double>(h, g, &g->front_inv, (size_t)pp::PP_WF * pp::PP_WF * d.bpad))) return rc;
    }
    for (int q = 0; q < 16; ++q) fent.insert(fent.end(), {0, 0, 0, 0});   // slack for the vector record reads
    // tile records (one per panel) -> column-step records (one per panel column), with their own tile pointers
    std::vector<int> sptr(P.stile_ptr.size(), 0);
    for (size_t tix = 0; tix + 1 < P.stile_ptr.size(); ++tix) {
      sptr[tix] = (int)(srec.size() / 20);
      for (int ri = P.stile_ptr[tix]; ri < P.stile_ptr[tix + 1]; ++ri) {
        const auto& r = P.stile_rec[ri];
        const int w = P.piv_w[r.piv];
        for (int t = 0; t < w; ++t) {
          srec.insert(srec.end(), {w, (int)(P.piv_uoff[r.piv] + t), r.piv, t});
          for (int q = 0; q < 8; ++q) srec.push_back(r.slotA[q]);
          for (int q = 0; q < 8; ++q) srec.push_back(r.slotB[q]);
        }
      }
    }
    if (!sptr.empty()) sptr.back() = (int)(srec.size() / 20);
    if ((rc = dev_upload(h, g, &d.piv_w, P.piv_w))) return rc;
    if ((rc = dev_upload(h, g, &d.piv_start, P.piv_start))) return rc;
    if ((rc = dev_upload(h, g, &d.piv_uoff, uoff))) return rc;
    if ((rc = dev_upload(h, g, &d.piv_doff, P.piv_doff))) return rc;
    {
      std::vector<int> sub(P.piv_sub.begin(), P.piv_sub.end());
      if ((rc = dev_upload(h, g, &d.piv_sub, sub))) return rc;
      if ((rc = dev_upload(h, g, &d.piv_boff, P.piv_boff))) return rc;
      if ((rc = dev_upload(h, g, &d.piv_of_col, P.piv_of_col))) return rc;
    }
    if ((rc = dev_upload(h, g, &d.piv_rowptr, P.piv_rowptr))) return rc;
    if ((rc = dev_upload(h, g, &d.rowidx, P.rowidx))) return rc;
    if ((rc = dev_upload(h, g, &d.perm, P.perm))) return rc;
    if ((rc = dev_upload(h, g, &d.iperm, P.iperm))) return rc;
    if ((rc = dev_upload(h, g, &d.rawmap, rawmap))) return rc;
    {
      std::vector<int> rtiles;
      for (int t0 = 0; t0 * 64 < g->nraw; ++t0) {
        bool any = false;
        for (int e = t0 * 64; e < std::min(g->nraw, t0 * 64 + 64) && !any; ++e) any = rawmap[(size_t)e] >= 0;
        if (any) rtiles.push_back(t0);
      }
      g->nraw_tiles = (int)rtiles.size();
      rtiles.push_back(0);
      if ((rc = dev_upload(h, g, &d.raw_tiles, rtiles))) return rc;
    }
    if ((rc = dev_upload(h, g, &d.ftask, ftask))) return rc;
    if ((rc = dev_upload(h, g, &d.stask, stask))) return rc;
    if ((rc = dev_upload(h, g, &d.fdst_ptr, fdst_ptr))) return rc;
    if ((rc = dev_upload(h, g, &d.fent, fent))) return rc;
    g->fent_host = fent;
    d.const_row = -1;
    if (!g->cmap_host.empty()) {      // [batch][nc_loc] -> [nc_loc][bpad] (padded lanes repeat instance 0: never used)
      std::vector<int> cm((size_t)std::max(g->nc_loc, 1) * d.bpad, 0);
      for (int c = 0; c < g->nc_loc; ++c)
        for (int b = 0; b < d.bpad; ++b)
          cm[(size_t)c * d.bpad + b] = g->cmap_host[(size_t)(b < g->batch ? b : 0) * g->nc_loc + c];
      if ((rc = dev_upload(h, g, &d.cmapT, cm))) return rc;
      d.xs_row = d.bpad; d.xs_lane = 1;
    }
    if ((rc = dev_upload(h, g, &d.clevel_col, P.clevel_col))) return rc;
    {
      std::vector<int> frec, brec;
      frec.reserve(P.clevel_col.size() * 4);
      brec.reserve(P.clevel_col.size() * 8);
      for (int c : P.clevel_col) {
        const int pv = P.piv_of_col[c], w = P.piv_w[pv], q = c - P.piv_start[pv];
        frec.insert(frec.end(), {c, P.perm[c], P.sfwd_eptr[c], P.sfwd_eptr[c + 1]});
        brec.insert(brec.end(), {c, w, q, P.piv_rowptr[pv + 1] - P.piv_rowptr[pv], P.piv_rowptr[pv],
                                 (int)(P.piv_uoff[pv] + (int64_t)w * w + q), P.piv_doff[pv], P.piv_start[pv]});
      }
      g->fwd_level_has_entries.assign((size_t)P.n_levels, 0);
      for (int l = 0; l < P.n_levels; ++l)
        for (int q = P.clevel_ptr[l]; q < P.clevel_ptr[l + 1]; ++q) {
          const int c = P.clevel_col[q];
          if (P.sfwd_eptr[c + 1] > P.sfwd_eptr[c]) { g->fwd_level_has_entries[(size_t)l] = 1; break; }
        }
      // wave teams: a row / column with more than a couple of 16-entry load rounds is shared by 4 or 16 waves
      // (thresholds measured at C3: 16/48 and 16/64 were slower, 48/128 the same)
      auto team_of = [](int longest) { return longest > 96 ? 16 : longest > 32 ? 4 : 1; };
      g->fwd_level_team.assign((size_t)P.n_levels, 1);
      g->bwd_level_team.assign((size_t)P.n_levels, 1);
      g->fwd_level_maxrow.assign((size_t)P.n_levels, 0);
      g->bwd_level_maxrow.assign((size_t)P.n_levels, 0);
      for (int l = 0; l < P.n_levels; ++l) {
        int fmax = 0, bmax = 0;
        for (int q = P.clevel_ptr[l]; q < P.clevel_ptr[l + 1]; ++q) {
          const int c = P.clevel_col[q], pv = P.piv_of_col[c];
          fmax = std::max(fmax, P.sfwd_eptr[c + 1] - P.sfwd_eptr[c]);
          bmax = std::max(bmax, P.piv_rowptr[pv + 1] - P.piv_rowptr[pv]);
        }
        g->fwd_level_team[(size_t)l] = team_of(fmax);
        g->bwd_level_team[(size_t)l] = team_of(bmax);
        g->fwd_level_maxrow[(size_t)l] = fmax;
        g->bwd_level_maxrow[(size_t)l] = bmax;
      }
      if ((rc = dev_upload(h, g, &d.fwd_rec, frec))) return rc;
      if ((rc = dev_upload(h, g, &d.bwd_rec, brec))) return rc;
      // native-vector variants: a column without incoming entries keeps y = b, which then is read from the caller's
      // right-hand side (row perm[c]) instead of a copy; x is written and read in the caller's row order
      std::vector<uint8_t> noent((size_t)P.n, 0);
      for (int c = 0; c < P.n; ++c) noent[(size_t)c] = P.sfwd_eptr[c + 1] == P.sfwd_eptr[c];
      std::vector<int> zf(P.sfwd_zcol), zc2(P.crow_zcol), brn(brec), ro(P.rowidx);
      for (auto& z : zf) if (noent[(size_t)z]) z = -1 - P.perm[z];
      for (auto& z : zc2) if (noent[(size_t)z]) z = -1 - P.perm[z];
      for (int q = 0; q < 16; ++q) { zf.push_back(0); zc2.push_back(0); }
      for (size_t i = 0; i < brn.size(); i += 8) {
        const int p0 = brn[i + 7], w = brn[i + 1];
        brn[i] = P.perm[brn[i]];
        // y of the whole block pivot is read from the right-hand side only if NONE of its columns has incoming entries
        // (then its level was not launched in the forward sweep); a level that was launched has written y for all of
        // its columns.  (Columns of one block pivot may differ: a panel below may hold only some of them as rows.)
        bool none = true;
        for (int t = 0; t < w; ++t) none = none && noent[(size_t)(p0 + t)];
        if (none) brn[i + 7] = -1 - p0;
      }
      for (auto& r : ro) if (r < P.n) r = P.perm[r];
      if ((rc = dev_upload(h, g, &g->zcolN_f, zf))) return rc;
      if ((rc = dev_upload(h, g, &g->zcolN_c, zc2))) return rc;
      if ((rc = dev_upload(h, g, &g->brecN, brn))) return rc;
      if ((rc = dev_upload(h, g, &g->rowidx_o, ro))) return rc;
    }
    {
      std::vector<int> up(P.sfwd_upos), zc(P.sfwd_zcol), cu(P.crow_upos), cz(P.crow_zcol);
      for (int q = 0; q < 16; ++q) { up.push_back(0); zc.push_back(0); cu.push_back(0); cz.push_back(0); }
      if ((rc = dev_upload(h, g, &d.sfwd_eptr, P.sfwd_eptr))) return rc;
      if ((rc = dev_upload(h, g, &d.sfwd_upos, up))) return rc;
      if ((rc = dev_upload(h, g, &d.sfwd_zcol, zc))) return rc;
      if ((rc = dev_upload(h, g, &d.crow_eptr, P.crow_eptr))) return rc;
      if ((rc = dev_upload(h, g, &d.crow_upos, cu))) return rc;
      if ((rc = dev_upload(h, g, &d.crow_zcol, cz))) return rc;
    }
    if ((rc = dev_upload(h, g, &d.stile_a, P.stile_a))) return rc;
    if ((rc = dev_upload(h, g, &d.stile_b, P.stile_b))) return rc;
    if ((rc = dev_upload(h, g, &d.stile_ptr, sptr))) return rc;
    if ((rc = dev_upload(h, g, &d.stile_rec, srec))) return rc;
    g->ntiles = (int)P.stile_a.size();
    // 16 x 16 tiles for the MFMA form (k_schur_mfma): per (tile pair, panel column) one record with the positions of the
    // 16 + 16 rows; the records of a tile are cut into work items of at most PP_MT_SLICE records
    g->mt_wide = h->nc >= PP_MT_WIDE_NC && std::getenv("PP_NO_WIDE_SCHUR_TILES") == nullptr;
    if (g->mt_wide) {
      // 32 x 32 super-tiles (k_schur_mfma_wide): per (super-tile pair, panel column) one record with the positions of the
      // 32 + 32 rows; mt_a / mt_b per quarter (4 per super-tile, -1: the quarter above the diagonal), items {r0, r1, super, 0}
      std::map<std::pair<int, int>, std::vector<int>> by_super;
      for (int pv = 0; pv < P.npiv; ++pv) {
        const int w = P.piv_w[pv];
        std::vector<int> sl;
        std::vector<std::array<int, 32>> slots;
        for (int q = P.piv_rowptr[pv]; q < P.piv_rowptr[pv + 1]; ++q) {
          const int r = P.rowidx[(size_t)q];
          if (r < P.n) continue;
          const int c = r - P.n, si = c / 32;
          if (sl.empty() || sl.back() != si) { sl.push_back(si); std::array<int, 32> e; e.fill(-1); slots.push_back(e); }
          slots.back()[(size_t)(c % 32)] = w + (q - P.piv_rowptr[pv]);
        }
        for (size_t a = 0; a < sl.size(); ++a)
          for (size_t b2 = 0; b2 <= a; ++b2)
            for (int t = 0; t < w; ++t) {
              auto& v = by_super[{sl[a], sl[b2]}];
              for (int q = 0; q < 32; ++q) v.push_back(slots[a][(size_t)q] < 0 ? -1 : (int)(P.piv_uoff[pv] + (int64_t)slots[a][(size_t)q] * w + t));
              for (int q = 0; q < 32; ++q) v.push_back(slots[b2][(size_t)q] < 0 ? -1 : (int)(P.piv_uoff[pv] + (int64_t)slots[b2][(size_t)q] * w + t));
            }
      }
      std::vector<int> mta, mtb, mrec, mitem, mwptr{0};
      int super = 0;
      for (auto& kv : by_super) {
        for (int x = 0; x < 2; ++x)
          for (int y = 0; y < 2; ++y) {
            const int ta = 2 * kv.first.first + x, tb = 2 * kv.first.second + y;
            mta.push_back(ta >= tb ? ta : -1);
            mtb.push_back(ta >= tb ? tb : -1);
          }
        const int r0 = (int)(mrec.size() / 64);
        mrec.insert(mrec.end(), kv.second.begin(), kv.second.end());
        const int r1 = (int)(mrec.size() / 64);
        for (int r = r0; r < r1; r += PP_MT_SLICE) mitem.insert(mitem.end(), {r, std::min(r1, r + PP_MT_SLICE), super, 0});
        mwptr.push_back((int)(mitem.size() / 4));
        ++super;
      }
      g->nmt = (int)mta.size();                       // quarters (workgroups of the reduction)
      g->nmt_items = (int)(mitem.size() / 4);
      if ((rc = dev_upload(h, g, &d.mt_a, mta))) return rc;
      if ((rc = dev_upload(h, g, &d.mt_b, mtb))) return rc;
      if ((rc = dev_upload(h, g, &d.mt_rec, mrec))) return rc;
      if ((rc = dev_upload(h, g, &d.mt_item, mitem))) return rc;
      if ((rc = dev_upload(h, g, &d.mt_wptr, mwptr))) return rc;
    } else {
      std::map<std::pair<int, int>, std::vector<int>> by_tile;      // (ta, tb) -> records of 32 ints
      for (int pv = 0; pv < P.npiv; ++pv) {
        const int w = P.piv_w[pv];
        std::vector<int> tl;
        std::vector<std::array<int, 16>> slots;
        for (int q = P.piv_rowptr[pv]; q < P.piv_rowptr[pv + 1]; ++q) {
          const int r = P.rowidx[(size_t)q];
          if (r < P.n) continue;
          const int c = r - P.n, ti = c / 16;
          if (tl.empty() || tl.back() != ti) { tl.push_back(ti); std::array<int, 16> e; e.fill(-1); slots.push_back(e); }
          slots.back()[(size_t)(c % 16)] = w + (q - P.piv_rowptr[pv]);      // row slot inside the panel
        }
        for (size_t a = 0; a < tl.size(); ++a)
          for (size_t b2 = 0; b2 <= a; ++b2)
            for (int t = 0; t < w; ++t) {
              auto& v = by_tile[{tl[a], tl[b2]}];
              for (int q = 0; q < 16; ++q) v.push_back(slots[a][(size_t)q] < 0 ? -1 : (int)(P.piv_uoff[pv] + (int64_t)slots[a][(size_t)q] * w + t));
              for (int q = 0; q < 16; ++q) v.push_back(slots[b2][(size_t)q] < 0 ? -1 : (int)(P.piv_uoff[pv] + (int64_t)slots[b2][(size_t)q] * w + t));
            }
      }
      std::vector<int> mta, mtb, mrec, mitem, mwptr{0};
      for (auto& kv : by_tile) {
        mta.push_back(kv.first.first); mtb.push_back(kv.first.second);
        const int r0 = (int)(mrec.size() / 32);
        mrec.insert(mrec.end(), kv.second.begin(), kv.second.end());
        const int r1 = (int)(mrec.size() / 32);
        for (int r = r0; r < r1; r += PP_MT_SLICE) { mitem.push_back(r); mitem.push_back(std::min(r1, r + PP_MT_SLICE)); }
        mwptr.push_back((int)(mitem.size() / 2));
      }
      g->nmt = (int)mta.size();
      g->nmt_items = (int)(mitem.size() / 2);
      if ((rc = dev_upload(h, g, &d.mt_a, mta))) return rc;
      if ((rc = dev_upload(h, g, &d.mt_b, mtb))) return rc;
      if ((rc = dev_upload(h, g, &d.mt_rec, mrec))) return rc;
      if ((rc = dev_upload(h, g, &d.mt_item, mitem))) return rc;
      if ((rc = dev_upload(h, g, &d.mt_wptr, mwptr))) return rc;
    }
  }
  int rc;
  if (h->btd && (h->G < 1 || h->gs < 1 || h->gs > 512 || (int64_t)h->G * h->gs != nc))
    return fail(h, 3, "block-tridiagonal coupling structure: need n_c = G * gs with 1 <= gs <= 512");
  for (Group* g : h->groups)
    if (h->btd && g->cmap_host.empty() && nc > 0) return fail(h, 3, "block-tridiagonal S needs mapped groups (pp_add_group_mapped)");
  const size_t nn = schur_doubles(h);
  const size_t nd = h->btd ? 1 : nn;            // the dense factor copies are not needed for a block-tridiagonal S
  if ((rc = dev_alloc<double>(h, nullptr, &h->S_own, nn + PP_TAIL))) return rc;
  if ((rc = dev_alloc<double>(h, nullptr, &h->Sfac, nd))) return rc;
  if ((rc = dev_alloc<double>(h, nullptr, &h->Sldl, nd))) return rc;
  if ((rc = dev_alloc<int>(h, nullptr, &h->scatter_err, 4))) return rc;
  PP_HIP(hipMemset(h->scatter_err, 0, 4 * sizeof(int)));
  if (h->btd) {
    const size_t g2 = (size_t)h->gs * h->gs;
    if ((rc = dev_alloc<double>(h, nullptr, &h->btd_fac, nn))) return rc;
    if ((rc = dev_alloc<double>(h, nullptr, &h->btd_inv, (size_t)h->G * g2))) return rc;
    if ((rc = dev_alloc<double>(h, nullptr, &h->btd_klo, (size_t)h->G * g2))) return rc;
    if ((rc = dev_alloc<double>(h, nullptr, &h->btd_kup, (size_t)h->G * g2))) return rc;
    if ((rc = dev_alloc<double>(h, nullptr, &h->btd_ylo, (size_t)h->G * g2))) return rc;
    if ((rc = dev_alloc<double>(h, nullptr, &h->btd_yup, (size_t)h->G * g2))) return rc;
    if ((rc = dev_alloc<double>(h, nullptr, &h->btd_vec, 8 * (size_t)nc + 64))) return rc;    // BK work | b | u
    if ((rc = dev_alloc<int>(h, nullptr, &h->btd_ipiv, nc))) return rc;
    if ((rc = dev_alloc<int>(h, nullptr, &h->btd_info, 4 * (size_t)h->G))) return rc;
    if ((rc = build_btd_schedule(h))) return rc;
  }
  if ((rc = dev_alloc<double>(h, nullptr, &h->dvec, nc))) return rc;
  if ((rc = dev_alloc<int>(h, nullptr, &h->dense_mode, 4))) return rc;
  PP_HIP(hipMemset(h->dense_mode, 0, 4 * sizeof(int)));
  // Q: dense n_c x n_c; for a block-tridiagonal S (the layout of S, tens of MB) only when a caller hands over a flat Q --
  // the sparse form of pp_factor_schur_corner needs none
  if (!h->btd && (rc = dev_alloc<double>(h, nullptr, &h->Qd, nn))) return rc;
  if ((rc = dev_alloc<double>(h, nullptr, &h->work, 2 * (size_t)nc))) return rc;
  if ((rc = dev_alloc<double>(h, nullptr, &h->rs_own, nc))) return rc;
  if ((rc = dev_alloc<double>(h, nullptr, &h->rcd, nc))) return rc;
  if ((rc = dev_alloc<double>(h, nullptr, &h->xc, nc))) return rc;
  if ((rc = dev_alloc<int>(h, nullptr, &h->ipiv, nc))) return rc;
  if ((rc = dev_alloc<int>(h, nullptr, &h->bkinfo, 4))) return rc;
  if ((rc = dev_alloc<int>(h, nullptr, &h->counters, 4 * PP_CSLOTS))) return rc;
  PP_HIP(hipMemset(h->counters, 0, 4 * PP_CSLOTS * sizeof(int)));     // (afterwards cleared by the kernel that writes the tail)
  {
    void* hp = nullptr;
    void* dp = nullptr;
    PP_HIP(hipHostMalloc(&hp, 8 * sizeof(long long), hipHostMallocMapped));
    std::memset(hp, 0, 8 * sizeof(long long));
    PP_HIP(hipHostGetDevicePointer(&dp, hp, 0));
    h->status_host = (volatile long long*)hp;
    h->status_dev = (long long*)dp;
    h->status_seq = 0;
  }
  h->S = h->S_own;
  h->rs = h->rs_own;
  PP_HIP(hipMemset(h->S, 0, (nn + PP_TAIL) * sizeof(double)));
  PP_HIP(hipMemset(h->rs, 0, std::max<size_t>(nc, 1) * sizeof(double)));
  PP_HIP(hipMemset(h->bkinfo, 0, 4 * sizeof(int)));
  h->symbolic_done = true;
  // value storage (factor panels, work vectors): sized by the plan; if it does not fit the handle's budget the
  // symbolic phase still succeeds (the plan is valid) and the numeric phase reports not_enough_memory until
  // increase_memory_allocation has raised the budget (reference: ma27_interface.py:126-131, 153-154)
  h->values_allocated = false;
  h->mem_required = value_storage_bytes(h);
  (void)alloc_value_storage(h);
  h->err.clear();
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
  if (int rc = alloc_value_storage(h)) return rc;
  if (!g->dev.raw) { if (int rc = ensure_optional(h, g, OPT_RAW)) return rc; }
  g->input_mode = Group::IN_RAW;
  const size_t bytes = (size_t)g->batch * g->nraw * sizeof(double);
  if (bytes == 0 || raw == g->dev.raw) return 0;
  PP_HIP(hipMemcpyAsync(g->dev.raw, raw, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
  return 0;
}

int pp_upload_values_compact(pp_handle h, int group, const double* compact, int row0, int nrows, int on_device) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_upload_values_compact: bad group or symbolic phase not finished");
  if (row0 < 0 || nrows < 0 || row0 + nrows > g->batch) return fail(h, 3, "pp_upload_values_compact: row range outside the batch");
  PP_HIP(hipSetDevice(h->device));
  if (int rc = ensure_optional(h, g, OPT_RAW)) return rc;
  g->input_mode = Group::IN_COMPACT;
  const size_t stride = (size_t)g->nraw_used;
  if (nrows == 0 || stride == 0) return 0;
  PP_HIP(hipMemcpyAsync(g->raw_own + (size_t)row0 * stride, compact + (size_t)row0 * stride, (size_t)nrows * stride * sizeof(double),
                        on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
  return 0;
}

int pp_used_raw_entries(pp_handle h, int group, int32_t* out, int capacity) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_used_raw_entries: bad group or symbolic phase not finished");
  if (capacity < (int)g->used_raw.size()) return fail(h, 3, "pp_used_raw_entries: buffer too small");
  std::memcpy(out, g->used_raw.data(), g->used_raw.size() * sizeof(int));
  return 0;
}

int pp_set_value_map(pp_handle h, int group, int nsrc, const int32_t* src_of_raw, const double* coef_of_raw) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_set_value_map: bad group or symbolic phase not finished");
  if (nsrc < 0 || !src_of_raw || !coef_of_raw) return fail(h, 3, "pp_set_value_map: bad arguments");
  PP_HIP(hipSetDevice(h->device));
  PP_HIP(hipStreamSynchronize(h->stream));
  std::vector<int> ms(g->used_raw.size() + 1, -1);
  std::vector<double> mc(g->used_raw.size() + 1, 0.0);
  for (size_t j = 0; j < g->used_raw.size(); ++j) {
    const int e = g->used_raw[j];
    if (src_of_raw[e] >= nsrc) return fail(h, 3, "pp_set_value_map: source row out of range");
    ms[j] = src_of_raw[e] < 0 ? -1 : src_of_raw[e];
    mc[j] = coef_of_raw[e];
  }
  for (void* p : {(void*)g->map_src, (void*)g->map_coef, (void*)g->src_own}) if (p) (void)hipFree(p);
  g->map_src = nullptr; g->map_coef = nullptr; g->src_own = nullptr; g->src = nullptr;
  g->nsrc = nsrc;
  int rc;
  if ((rc = dev_alloc(h, (Group*)nullptr, &g->map_src, ms.size()))) return rc;
  if ((rc = dev_alloc(h, (Group*)nullptr, &g->map_coef, mc.size()))) return rc;
  PP_HIP(hipMemcpy(g->map_src, ms.data(), ms.size() * sizeof(int), hipMemcpyHostToDevice));
  PP_HIP(hipMemcpy(g->map_coef, mc.data(), mc.size() * sizeof(double), hipMemcpyHostToDevice));
  // entry records for the fused path: an initial-value record reads its source row directly (row nsrc = the constant
  // 1) and carries its coefficient, so the factorisation kernels gather from the sources themselves
  {
    std::vector<int> fs(g->fent_host);
    for (int pos : g->init_rec) {
      const int row = -1 - fs[(size_t)4 * pos];            // compact row of the transposed input
      const int sidx = ms[(size_t)row];
      const double c = mc[(size_t)row];
      int bits[2];
      std::memcpy(bits, &c, sizeof(c));
      fs[(size_t)4 * pos] = -1 - (sidx >= 0 ? sidx : nsrc);
      fs[(size_t)4 * pos + 1] = bits[0];
      fs[(size_t)4 * pos + 2] = bits[1];
    }
    if (g->fent_src) { (void)hipFree(g->fent_src); g->fent_src = nullptr; }
    if ((rc = dev_alloc(h, (Group*)nullptr, &g->fent_src, fs.size()))) return rc;
    PP_HIP(hipMemcpy(g->fent_src, fs.data(), fs.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  return 0;
}

double* pp_source_buffer(pp_handle h, int group) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done || !g->map_src) return nullptr;
  if (!g->src_own) {
    if (dev_alloc(h, (Group*)nullptr, &g->src_own, (size_t)std::max(g->nsrc, 1) * (size_t)g->dev.bpad)) return nullptr;
    if (hipMemset(g->src_own, 0, (size_t)std::max(g->nsrc, 1) * (size_t)g->dev.bpad * sizeof(double)) != hipSuccess) return nullptr;
  }
  if (!g->src) g->src = g->src_own;
  return g->src_own;
}

int pp_bind_source_buffer(pp_handle h, int group, double* dev_ptr) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done || !g->map_src) return fail(h, 3, "pp_bind_source_buffer: bad group or no value map");
  if (!dev_ptr && !pp_source_buffer(h, group)) return fail(h, 1, "pp_bind_source_buffer: could not allocate the source buffer");
  g->src = dev_ptr ? dev_ptr : g->src_own;
  g->input_mode = Group::IN_SOURCES;
  return 0;
}

int pp_upload_sources(pp_handle h, int group, const double* src, int on_device) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done || !g->map_src) return fail(h, 3, "pp_upload_sources: bad group or no value map");
  PP_HIP(hipSetDevice(h->device));
  if (int rc = ensure_optional(h, g, OPT_RAW)) return rc;       // (host sources are staged through the raw buffer)
  if (!pp_source_buffer(h, group)) return fail(h, 1, "pp_upload_sources: could not allocate the source buffer");
  // [batch][nsrc] (one row per block, the producer's natural layout on the host) -> [nsrc][bpad]: staged through the
  // raw buffer (nsrc <= nraw is not required: the copy is done in slabs of whole rows)
  const size_t per = (size_t)g->nsrc;
  if (per == 0) { g->src = g->src_own; g->input_mode = Group::IN_SOURCES; return 0; }
  if (per > (size_t)g->nraw) return fail(h, 3, "pp_upload_sources: more sources than raw entries per block");
  PP_HIP(hipMemcpyAsync(g->raw_own, src, (size_t)g->batch * per * sizeof(double),
                        on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_transpose_in, dim3((unsigned)((g->nsrc + 63) / 64) * g->dev.nchunk), dim3(256), 0, h->stream, g->raw_own,
                     g->src_own, (const int*)nullptr, g->batch, g->nsrc, g->dev.bpad, 1, (const int*)nullptr);
  PP_HIP(hipGetLastError());
  g->src = g->src_own;
  g->input_mode = Group::IN_SOURCES;
  return 0;
}

double* pp_raw_buffer(pp_handle h, int group) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done || alloc_value_storage(h)) return nullptr;
  if (!g->dev.raw && ensure_optional(h, g, OPT_RAW)) return nullptr;
  return g->dev.raw;
}

int pp_numeric_factor_blocks(pp_handle h) {
  if (!h || !h->symbolic_done) return fail(h, 3, "pp_numeric_factor_blocks before symbolic factorization");
  PP_HIP(hipSetDevice(h->device));
  if (int rc = alloc_value_storage(h)) return rc;
  GroupStreams gst;
  if (fork_group_streams(h, gst)) return fail(h, 3, "stream fork failed");
  auto group_body = [&](size_t gi) -> int {
    Group* g = h->groups[gi];
    const hipStream_t st = gst.st[gi % (size_t)gst.n];
    const pp::Plan& P = g->plan;
    GroupDev& d0 = g->dev;
    bool fused_sources = false;
    d0.lbound = h->growth_bound > 0.0 ? h->growth_bound : INFINITY;
    const bool shifting = g->nshift > 0 && (h->shift_w != 0.0 || h->shift_c != 0.0);
    fused_sources = g->input_mode == Group::IN_SOURCES && g->fent_src && !shifting && !h->no_fused_sources;
    if (!fused_sources) {       // the transposed input exists only for the paths that assemble into it
      if (int rc = ensure_optional(h, g, OPT_RAWT)) return rc;
      if (g->input_mode == Group::IN_RAW && d0.nraw > 0 && !d0.raw)
        return fail(h, 3, "pp_numeric_factor_blocks: no values uploaded");
      if (g->input_mode == Group::IN_COMPACT && g->nraw_used > 0 && !g->raw_own)
        return fail(h, 3, "pp_numeric_factor_blocks: no values uploaded");
    }
    GroupDev d = d0;
    {
      PhaseScope ps(h, 0, fused_sources ? 0 : 1);
      if (g->input_mode == Group::IN_SOURCES && (!g->src || !g->map_src))
        return fail(h, 3, "pp_numeric_factor_blocks: no value map / source buffer");
      if (fused_sources) {
        // nothing to assemble: the factorisation kernels read the sources through the entry records
      } else if (g->input_mode == Group::IN_SOURCES && g->nraw_used > 0) {
        hipLaunchKernelGGL(k_assemble_sources, dim3((unsigned)((g->nraw_used + 3) / 4) * d.nchunk), dim3(256), 0, st, g->src,
                           d.rawT, g->map_src, g->map_coef, g->nraw_used, d.bpad);
      } else if (g->input_mode == Group::IN_COMPACT && g->nraw_used > 0) {
        hipLaunchKernelGGL(k_transpose_in, dim3((unsigned)((g->nraw_used + 63) / 64) * d.nchunk), dim3(256), 0, st, g->raw_own,
                           d.rawT, (const int*)nullptr, d.batch, g->nraw_used, d.bpad, 1, (const int*)nullptr);
      } else if (d.nraw > 0)
      {
        const int tiles = transpose_tiles(d.nraw, d.nchunk);
        if (tiles == 1 && g->nraw_tiles > 0)
          hipLaunchKernelGGL(k_transpose_in, dim3((unsigned)g->nraw_tiles * d.nchunk), dim3(256), 0, st, d.raw, d.rawT, d.rawmap,
                             d.batch, d.nraw, d.bpad, 1, d.raw_tiles);
        else
          hipLaunchKernelGGL(k_transpose_in, dim3((unsigned)((d.nraw + 64 * tiles - 1) / (64 * tiles)) * d.nchunk), dim3(256), 0, st,
                             d.raw, d.rawT, d.rawmap, d.batch, d.nraw, d.bpad, tiles, (const int*)nullptr);
      }
      if (g->nshift > 0 && (h->shift_w != 0.0 || h->shift_c != 0.0))
        hipLaunchKernelGGL(k_shift_diag, dim3(g->nshift, (d.bpad + 255) / 256), dim3(256), 0, st, d.rawT, g->shift_row,
                           g->shift_cls, g->nshift, d.bpad, h->shift_w, h->shift_c);
    }
    if (fused_sources) { d.rawT = g->src; d.fent = g->fent_src; d.const_row = g->nsrc; }
    {
      int nlaunch = 0;
      for (int l = 0; l < P.n_levels; ++l)
        nlaunch += (P.flevel_ptr[l + 1] > P.flevel_ptr[l]) + (P.slevel_ptr[l + 1] > P.slevel_ptr[l]);
      if (P.front_piv >= 0) nlaunch += 1 + (P.wtasks.empty() ? 0 : 1);
      PhaseScope ps(h, 1, nlaunch);
      const Splits sp = make_splits(h, d.nchunk);
      hipStream_t fan[PP_MAX_SPLIT];
      if (fork_streams(h, sp, fan, st)) return fail(h, 3, "stream fork failed");
      for (int l = 0; l < P.n_levels; ++l) {
        const int t0 = P.flevel_ptr[l], nt = P.flevel_ptr[l + 1] - t0;
        const int s0 = P.slevel_ptr[l], ns = P.slevel_ptr[l + 1] - s0;
        for (int q = 0; q < sp.n; ++q) {
          const int ny = sp.c0[q + 1] - sp.c0[q];
          if (nt > 0) {
            const bool lean = P.flevel_maxent[l] <= 12 && P.flevel_nsplit[l] == 0;      // (the lean kernel has no split rows)
            const int mw = g->level_maxw[l];
            // two instances per lane where the chunks pair up (chunk counts and offsets in units of 128 instances)
            const bool pair = h->lane_pairs && ny % 2 == 0 && sp.c0[q] % 2 == 0;
#define PP_LAUNCH_FLAT(WM) do { \
              if (pair) hipLaunchKernelGGL((k_gather_flat<WM, 1, 2>), dim3((unsigned)nt * (ny / 2)), dim3(64), 0, fan[q], d, t0, sp.c0[q] / 2, ny / 2, PIVOT_EPS); \
              else hipLaunchKernelGGL((k_gather_flat<WM, 1, 1>), dim3((unsigned)nt * ny), dim3(64), 0, fan[q], d, t0, sp.c0[q], ny, PIVOT_EPS); } while (0)
#define PP_LAUNCH_FLAT_QUADS(WM) do { \
              if (pair) hipLaunchKernelGGL((k_gather_flat<WM, PP_QUAD, 2>), dim3((unsigned)(nt / PP_QUAD) * (ny / 2)), dim3(64 * PP_QUAD), 0, fan[q], d, t0, sp.c0[q] / 2, ny / 2, PIVOT_EPS); \
              else hipLaunchKernelGGL((k_gather_flat<WM, PP_QUAD, 1>), dim3((unsigned)(nt / PP_QUAD) * ny), dim3(64 * PP_QUAD), 0, fan[q], d, t0, sp.c0[q], ny, PIVOT_EPS); } while (0)
            if (lean) {
#define PP_LAUNCH_LEAN(WM) do { \
              if (pair) hipLaunchKernelGGL((k_gather_level_lean<WM, 2>), dim3((unsigned)nt * (ny / 2)), dim3(64), 0, fan[q], d, t0, sp.c0[q] / 2, ny / 2, PIVOT_EPS); \
              else hipLaunchKernelGGL((k_gather_level_lean<WM, 1>), dim3((unsigned)nt * ny), dim3(64), 0, fan[q], d, t0, sp.c0[q], ny, PIVOT_EPS); } while (0)
              if (mw == 1) PP_LAUNCH_LEAN(1);
              else if (mw == 2) PP_LAUNCH_LEAN(2);
              else if (mw <= 4) PP_LAUNCH_LEAN(4);
              else PP_LAUNCH_LEAN(PP_WMAX);
#undef PP_LAUNCH_LEAN
            }
            else if (P.flevel_nsplit[l] == 0) {
              if (mw == 1) PP_LAUNCH_FLAT(1);
              else if (mw == 2) PP_LAUNCH_FLAT(2);
              else if (mw <= 4) PP_LAUNCH_FLAT(4);
              else PP_LAUNCH_FLAT(PP_WMAX);
            } else {
              if (mw == 1) PP_LAUNCH_FLAT_QUADS(1);
              else if (mw == 2) PP_LAUNCH_FLAT_QUADS(2);
              else if (mw <= 4) PP_LAUNCH_FLAT_QUADS(4);
              else PP_LAUNCH_FLAT_QUADS(PP_WMAX);
            }
#undef PP_LAUNCH_FLAT
#undef PP_LAUNCH_FLAT_QUADS
          }
          if (ns > 0) {
            if (g->level_maxw[l] <= 4)
              hipLaunchKernelGGL(k_scale_level<4>, dim3((unsigned)ns * ny), dim3(64), 0, fan[q], d, s0, sp.c0[q], ny, PIVOT_EPS);
            else
              hipLaunchKernelGGL(k_scale_level<PP_WMAX>, dim3((unsigned)ns * ny), dim3(64), 0, fan[q], d, s0, sp.c0[q], ny, PIVOT_EPS);
          }
        }
        if (P.front_piv >= 0 && P.piv_level[P.front_piv] == l) {
          // root front: pivot block inverted by one workgroup per chunk, rows scaled with the explicit inverse
          if (sp.n != 1) return fail(h, 3, "instance splits are not supported together with a root front");
          const int fp = P.front_piv;
          const FrontRec fr = {fp, P.piv_w[fp], (int)P.piv_uoff[fp], P.piv_boff[fp], P.piv_doff[fp], P.piv_sub[fp]};
          hipLaunchKernelGGL(k_front_invert, dim3((unsigned)d.nchunk), dim3(512), 0, fan[0], d, fr, g->front_inv, PIVOT_EPS);
          if (!P.wtasks.empty())
            hipLaunchKernelGGL(k_scale_wide, dim3((unsigned)P.wtasks.size() * d.nchunk), dim3(192), 0, fan[0], d, g->wtask, fr,
                               g->front_inv, d.nchunk);
        }
      }
      if (join_streams(h, sp, fan)) return fail(h, 3, "stream join failed");
    }
    return 0;
  };
  if (int rc = run_groups(h, gst, group_body)) return rc;
  if (join_group_streams(h, gst)) return fail(h, 3, "stream join failed");
  PP_HIP(hipGetLastError());
  h->blocks_factored = true;
  h->numeric_done = false;
  h->schur_done = false;
  return 0;
}

int pp_numeric_schur(pp_handle h) {
  if (!h || !h->symbolic_done || !h->blocks_factored) return fail(h, 3, "pp_numeric_schur before pp_numeric_factor_blocks");
  PP_HIP(hipSetDevice(h->device));
  hipStream_t st = h->stream;
  const int nc = h->nc;
  // S starts from zero -- unless the first group is a plain (unmapped) one whose tiles cover all of S: its reduction
  // then stores instead of adding.  The counters are cleared by whoever writes the tail (zero at allocation).
  bool first_covers = false;
  if (!h->groups.empty() && !h->btd) {
    const Group* g0 = h->groups.front();
    const int nt8 = (nc + 7) / 8;
    first_covers = g0->ntiles > 0 && !g0->dev.cmapT && g0->ntiles == nt8 * (nt8 + 1) / 2;
  }
  if (!first_covers) PP_HIP(hipMemsetAsync(h->S, 0, (schur_doubles(h) + PP_TAIL) * sizeof(double), st));
  bool tail_written = false;
  for (Group* g : h->groups) {
    const pp::Plan& P = g->plan;
    GroupDev& d = g->dev;
    {
      PhaseScope ps(h, 2, g->ntiles > 0 ? 2 : 1);
      const size_t total8 = (size_t)P.npiv * d.bpad / 8;   // bpad is a multiple of 64
      const unsigned ncb = (unsigned)std::min<size_t>(2048, (total8 + 255) / 256);   // counting workgroups in front of the tiles
      if (g->ntiles > 0 && d.cmapT) {
        // mapped group: per-instance cliques, scattered into the dense or the block-tridiagonal S
        hipLaunchKernelGGL(k_schur_tiles, dim3((unsigned)g->ntiles * d.nchunk + ncb, 1, 2), dim3(64), 0, st, d, g->ntiles, total8,
                           h->counters);
        const SchurTarget T{h->S, nc, h->btd, h->gs, h->G, h->scatter_err};
        hipLaunchKernelGGL(k_scatter_schur, dim3((unsigned)g->ntiles * d.nchunk), dim3(64), 0, st, d, g->ntiles, T);
      } else if (g->ntiles > 0 && h->schur_mfma && g->nmt > 0) {
        if (g->mt_wide)
          hipLaunchKernelGGL(k_schur_mfma_wide, dim3((unsigned)g->nmt_items * d.nchunk + ncb), dim3(64), 0, st, d, g->nmt_items, total8,
                             h->counters);
        else
          hipLaunchKernelGGL(k_schur_mfma, dim3((unsigned)g->nmt_items * d.nchunk + ncb), dim3(64), 0, st, d, g->nmt_items, total8,
                             h->counters);
        const bool last = (g == h->groups.back());
        hipLaunchKernelGGL(k_schur_reduce_mfma, dim3(g->nmt), dim3(1024), 0, st, d, g->nmt_items, h->S,
                           last ? h->counters : (int*)nullptr, (first_covers && g == h->groups.front()) ? 1 : 0, g->mt_wide ? 1 : 0);
        tail_written = last;
      } else if (g->ntiles > 0) {
        hipLaunchKernelGGL(k_schur_tiles, dim3((unsigned)g->ntiles * d.nchunk + ncb, 1, 2), dim3(64), 0, st, d, g->ntiles, total8,
                           h->counters);
        const bool last = (g == h->groups.back());
        hipLaunchKernelGGL(k_schur_reduce, dim3(g->ntiles), dim3(64), 0, st, d, g->ntiles, h->S,
                           last ? h->counters : (int*)nullptr, (first_covers && g == h->groups.front()) ? 1 : 0);
        tail_written = last;
      } else {
        hipLaunchKernelGGL(k_count_codes, dim3((unsigned)std::min<size_t>(512, (total8 + 255) / 256)), dim3(256), 0, st,
                           d.codes, total8, h->counters, d.growth, d.growth + d.bpad, d.batch);
      }
    }
  }
  if (!tail_written) hipLaunchKernelGGL(k_write_tail, dim3(1), dim3(64), 0, st, h->counters, h->S + schur_doubles(h));
  PP_HIP(hipGetLastError());
  h->numeric_done = true;
  h->schur_done = false;
  return 0;
}

int pp_numeric_local(pp_handle h) {
  if (int rc = pp_numeric_factor_blocks(h)) return rc;
  return pp_numeric_schur(h);
}

int pp_fail_local(pp_handle h, int status) {
  if (!h || !h->symbolic_done) return fail(h, 3, "pp_fail_local before symbolic factorization");
  if (status < 1 || status > 3) return fail(h, 3, "pp_fail_local: status must be 1 (not_enough_memory), 2 (singular) or 3 (error)");
  PP_HIP(hipSetDevice(h->device));
  const size_t nn = schur_doubles(h);
  PP_HIP(hipMemsetAsync(h->S, 0, (nn + PP_TAIL) * sizeof(double), h->stream));
  h->fail_code = status == 1 ? 1.0 : status == 2 ? 1e3 : 1e6;
  PP_HIP(hipMemcpyAsync(h->S + nn + 3, &h->fail_code, sizeof(double), hipMemcpyHostToDevice, h->stream));
  h->numeric_done = true;       // the Schur buffer is defined (zero contribution): the collective and the dense phase may run
  h->schur_done = false;
  return 0;
}

double* pp_schur_buffer(pp_handle h) { return (h && h->symbolic_done) ? h->S : nullptr; }

int pp_bind_schur_buffer(pp_handle h, double* dev_ptr) {
  if (!h || !h->symbolic_done) return fail(h, 3, "pp_bind_schur_buffer before symbolic factorization");
  h->S = dev_ptr ? dev_ptr : h->S_own;
  return 0;
}

static int factor_schur_impl(pp_handle h, const double* Q_host, long long corner_nnz);

int pp_factor_schur(pp_handle h, const double* Q_host) { return factor_schur_impl(h, Q_host, 0); }

int pp_factor_schur_corner(pp_handle h, int64_t nnz, const int64_t* pos, const double* val) {
  if (!h || !h->numeric_done) return fail(h, 3, "pp_factor_schur_corner before pp_numeric_local");
  if (!h->btd) return fail(h, 3, "pp_factor_schur_corner: S is dense (use pp_factor_schur)");
  if (nnz < 0 || (nnz > 0 && (!pos || !val))) return fail(h, 3, "pp_factor_schur_corner: bad arguments");
  PP_HIP(hipSetDevice(h->device));
  const long long nn = (long long)schur_doubles(h);
  for (int64_t k = 0; k < nnz; ++k)
    if (pos[k] < 0 || pos[k] >= nn) return fail(h, 3, "pp_factor_schur_corner: position outside the Schur buffer");
  if (nnz > 0) {
    // the pairs travel on an upload stream of their own: a copy on the handle's stream would wait behind the block
    // factorisation that is still running there, and the host with it
    if (!h->ev_corner_up) {
      PP_HIP(hipStreamCreateWithFlags(&h->up_stream, hipStreamNonBlocking));
      PP_HIP(hipEventCreateWithFlags(&h->ev_corner_up, hipEventDisableTiming));
      PP_HIP(hipEventCreateWithFlags(&h->ev_corner_done, hipEventDisableTiming));
    }
    hipStream_t up = h->up_stream;
    if ((size_t)nnz > h->corner_cap) {
      PP_HIP(hipStreamSynchronize(h->stream));
      if (h->corner_pos) (void)hipFree(h->corner_pos);
      if (h->corner_val) (void)hipFree(h->corner_val);
      h->corner_pos = nullptr; h->corner_val = nullptr; h->corner_cap = 0;
      int rc = 0;
      if ((rc = dev_alloc<long long>(h, nullptr, &h->corner_pos, (size_t)nnz))) return rc;
      if ((rc = dev_alloc<double>(h, nullptr, &h->corner_val, (size_t)nnz))) return rc;
      h->corner_cap = (size_t)nnz;
      h->corner_used = false;
    }
    if (h->corner_used) PP_HIP(hipStreamWaitEvent(up, h->ev_corner_done, 0));    // the previous scatter has read them
    PP_HIP(hipMemcpyAsync(h->corner_pos, pos, (size_t)nnz * sizeof(long long), hipMemcpyHostToDevice, up));
    PP_HIP(hipMemcpyAsync(h->corner_val, val, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice, up));
    PP_HIP(hipEventRecord(h->ev_corner_up, up));
    PP_HIP(hipStreamSynchronize(up));            // the caller's arrays are free again when this returns
    PP_HIP(hipStreamWaitEvent(h->stream, h->ev_corner_up, 0));
  }
  return factor_schur_impl(h, nullptr, (long long)nnz);
}

static int factor_schur_impl(pp_handle h, const double* Q_host, long long corner_nnz) {
  if (!h || !h->numeric_done) return fail(h, 3, "pp_factor_schur before pp_numeric_local");
  PP_HIP(hipSetDevice(h->device));
  hipStream_t st = h->stream;
  const int nc = h->nc;
  const size_t nn = schur_doubles(h);
  if (nc > 0 && h->btd) {
    // block-tridiagonal S: sequential block LDL^T, Bunch-Kaufman inside the blocks (see k_btd_*)
    if (Q_host && !h->Qd) {
      if (int rc = dev_alloc<double>(h, nullptr, &h->Qd, nn)) return rc;
    }
    if (Q_host) PP_HIP(hipMemcpyAsync(h->Qd, Q_host, nn * sizeof(double), hipMemcpyHostToDevice, st));
    const int gs = h->gs, G = h->G;
    const size_t g2 = (size_t)gs * gs;
    const int nlev = (int)h->bcr_ne.size();
    PhaseScope ps(h, 3, 2 + 6 * nlev);
    hipLaunchKernelGGL(k_btd_init, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, nn, h->S, Q_host ? h->Qd : (const double*)nullptr,
                       h->btd_fac);
    if (corner_nnz > 0) {
      hipLaunchKernelGGL(k_corner_add, dim3((unsigned)((corner_nnz + 255) / 256)), dim3(256), 0, st, corner_nnz, h->corner_pos,
                         h->corner_val, h->btd_fac);
      PP_HIP(hipEventRecord(h->ev_corner_done, st));
      h->corner_used = true;
    }
    double* D = h->btd_fac;
    double* slot = h->btd_fac + (size_t)G * g2;
    const unsigned gb = (unsigned)((g2 + 255) / 256);
    // the diagonal blocks are factorised in LDS when they fit (gs <= 137: 150 KB of the CU's 160 KB)
    size_t lds_bytes = (g2 + 3 * (size_t)gs) * sizeof(double);      // the block, the two work columns of a 2 x 2 pivot step, the pivot indices
    if (lds_bytes > 150 * 1024) lds_bytes = 0;
    if (lds_bytes > 64 * 1024 && !h->bcr_lds_attr) {
      if (hipFuncSetAttribute((const void*)k_bcr_factor, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
        (void)hipGetLastError();
        lds_bytes = 0;
      } else {
        h->bcr_lds_attr = true;
      }
    }
    if (std::getenv("PP_BCR_NO_LDS")) lds_bytes = 0;
    const bool bcr_mfma = std::getenv("PP_NO_BCR_MFMA") == nullptr;    // block products on the matrix cores, wave-level inverse (measurement switch)
    // unpivoted LDL^T + inverse of the blocks on the matrix cores, Bunch-Kaufman only for the blocks it rejects
    // (PP_NO_BCR_LDL: measurement switch; needs the matrix-core products for the rest of the level and gs <= 112)
    bool bcr_ldl = bcr_mfma && gs <= 16 * BL_NT && std::getenv("PP_NO_BCR_LDL") == nullptr;
    if (bcr_ldl && !h->bcr_ldl_attr) {
      if (hipFuncSetAttribute((const void*)k_bcr_ldl_inverse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BL_LDS_BYTES) != hipSuccess) {
        (void)hipGetLastError();
        bcr_ldl = false;
      } else {
        h->bcr_ldl_attr = true;
      }
    }
    int bk_threads = 256;        // (measured at C4, gs = 98, S phase per step: 64 threads 15.3 ms, 128 12.1, 256 10.8, 512 11.0)
    if (const char* e = std::getenv("PP_BCR_THREADS")) bk_threads = std::max(64, std::min(BK_THREADS, std::atoi(e)));
    for (int l = 0; l < nlev; ++l) {
      const BcrLevel lv{h->btd_elim + h->bcr_off[(size_t)l], h->bcr_ne[(size_t)l], h->bcr_s[(size_t)l], h->bcr_lo[(size_t)l]};
      if (bcr_ldl)
        hipLaunchKernelGGL(k_bcr_ldl_inverse, dim3(lv.ne), dim3(BL_THREADS), BL_LDS_BYTES, st, gs, lv, D, h->btd_inv, h->btd_info, BK_EPS,
                           h->bcr_lbound);
      hipLaunchKernelGGL(k_bcr_factor, dim3(lv.ne), dim3(bk_threads), lds_bytes, st, gs, lv, D, h->btd_ipiv, h->btd_vec, h->btd_info,
                         lds_bytes > 0 ? 1 : 0, bcr_ldl ? 1 : 0);
      if (bcr_mfma) hipLaunchKernelGGL(k_bcr_invert_wave, dim3((gs + 3) / 4, lv.ne), dim3(256), 4 * (size_t)gs * sizeof(double), st, gs, lv, D,
                                       h->btd_ipiv, h->btd_inv, bcr_ldl ? h->btd_info : (const int*)nullptr);
      else hipLaunchKernelGGL(k_bcr_invert, dim3(gs, lv.ne), dim3(128), 0, st, gs, lv, D, h->btd_ipiv, h->btd_inv);
      if (l + 1 < nlev) {
        if (bcr_mfma) {
          const unsigned nt16 = (unsigned)((gs + 15) / 16);
          hipLaunchKernelGGL(k_bcr_keep_y_mfma, dim3(nt16 * nt16, lv.ne, 2), dim3(64), 0, st, gs, G, lv, h->btd_inv, slot, h->btd_klo,
                             h->btd_kup, h->btd_ylo, h->btd_yup);
          // (z = 0 and z = 2 of different eliminated blocks never meet: block i - s of one is block i + s of another only
          // across levels; within a level D_j is updated from below by z = 2 of i = j - s and from above by z = 0 of
          // i = j + s -- two read-modify-writes of the same block: two launches)
          hipLaunchKernelGGL(k_bcr_update_mfma, dim3(nt16 * nt16, lv.ne, 2), dim3(64), 0, st, gs, G, lv, h->btd_klo, h->btd_kup,
                             h->btd_ylo, h->btd_yup, D, slot, 0);
          hipLaunchKernelGGL(k_bcr_update_mfma, dim3(nt16 * nt16, lv.ne, 1), dim3(64), 0, st, gs, G, lv, h->btd_klo, h->btd_kup,
                             h->btd_ylo, h->btd_yup, D, slot, 2);
        } else {
        hipLaunchKernelGGL(k_bcr_keep_y, dim3(gb, lv.ne), dim3(256), 0, st, gs, G, lv, h->btd_inv, slot, h->btd_klo, h->btd_kup,
                           h->btd_ylo, h->btd_yup);
        hipLaunchKernelGGL(k_bcr_update, dim3(gb, lv.ne), dim3(256), 0, st, gs, G, lv, 0, h->btd_klo, h->btd_kup, h->btd_ylo,
                           h->btd_yup, D, slot);
        hipLaunchKernelGGL(k_bcr_update, dim3(gb, lv.ne), dim3(256), 0, st, gs, G, lv, 1, h->btd_klo, h->btd_kup, h->btd_ylo,
                           h->btd_yup, D, slot);
        }
      }
    }
    hipLaunchKernelGGL(k_btd_finish, dim3(1), dim3(256), 0, st, G, h->btd_info, h->bkinfo, h->S + nn, h->scatter_err, h->status_dev,
                       ++h->status_seq);
    PP_HIP(hipGetLastError());
    h->schur_done = true;
    return 0;
  }
  if (nc > 0) {
    if (Q_host) PP_HIP(hipMemcpyAsync(h->Qd, Q_host, nn * sizeof(double), hipMemcpyHostToDevice, st));
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
        // large S: panel + trailing update spread over the chip, two launches per 32 columns
        double* anorm = h->work;                    // (scratch of the Bunch-Kaufman fallback, free until then:
        double* stage = h->work + 8;                //  2 n_c doubles >= 8 + 32 * 32 for n_c > 512)
        int* flags = h->dense_mode + 2;
        hipLaunchKernelGGL(k_dense_anorm, dim3(1), dim3(256), 0, st, nc, h->Sldl, anorm, flags);
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
        hipLaunchKernelGGL(k_dense_finish, dim3(1), dim3(64), 0, st, nc, flags, h->dense_mode, h->bkinfo);
      }
    else
      PP_HIP(hipMemsetAsync(h->dense_mode, 0, sizeof(int), st));
    // Bunch-Kaufman on S + Q if the unpivoted factorisation was not accepted; publishes the status either way
    hipLaunchKernelGGL(k_bk_factor, dim3(1), dim3(BK_THREADS), 0, st, nc, h->S, Qd, h->Sfac, h->ipiv, h->work, h->bkinfo,
                       h->dense_mode, h->status_dev, ++h->status_seq);
  } else {
    PP_HIP(hipMemsetAsync(h->bkinfo, 0, 4 * sizeof(int), st));
    hipLaunchKernelGGL(k_publish_status, dim3(1), dim3(64), 0, st, h->S + nn, h->bkinfo, h->status_dev, ++h->status_seq);
  }
  PP_HIP(hipGetLastError());
  h->schur_done = true;
  return 0;
}

int pp_get_status(pp_handle h, int64_t out[4]) {
  if (!h || !h->schur_done) return fail(h, 3, "pp_get_status before pp_factor_schur");
  PP_HIP(hipSetDevice(h->device));
  // poll the mailbox; after a bounded spin fall back to a stream synchronisation (which also surfaces
  // an asynchronous device error instead of spinning on it)
  const long long want = h->status_seq;
  bool seen = false;
  const auto t0 = std::chrono::steady_clock::now();
  for (long spin = 0;; ++spin) {
    if (__atomic_load_n((const long long*)(h->status_host + 4), __ATOMIC_ACQUIRE) == want) { seen = true; break; }
    if ((spin & 1023) == 1023 &&
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.25) break;
  }
  if (!seen) {
    PP_HIP(hipStreamSynchronize(h->stream));
    if (__atomic_load_n((const long long*)(h->status_host + 4), __ATOMIC_ACQUIRE) != want)
      return fail(h, 3, "pp_get_status: status mailbox was not written");
  }
  for (int i = 0; i < 4; ++i) out[i] = (int64_t)h->status_host[i];
  // element growth beyond 1 / u_runtime, if the caller asked for the guard (pp_set_pivot_tolerance), is reported like a
  // breakdown: the host class refreshes the static pivot order from the offending instance and, if that does not help,
  // the inertia-correction loop regularises (MA27 would have re-pivoted).  Every rank sees the same all-reduced count.
  if (h->growth_fatal && out[0] == 0 && h->status_host[5] > 0) out[0] = 2;
  return 0;
}

int pp_get_schur(pp_handle h, double* S_host) {
  if (!h || !h->numeric_done) return fail(h, 3, "pp_get_schur before pp_numeric_local");
  PP_HIP(hipSetDevice(h->device));
  PP_HIP(hipMemcpyAsync(S_host, h->S, schur_doubles(h) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  PP_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

int pp_upload_rhs(pp_handle h, int group, const double* rhs, int on_device) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_upload_rhs: bad group or symbolic phase not finished");
  PP_HIP(hipSetDevice(h->device));
  if (int rc = alloc_value_storage(h)) return rc;
  if (!g->dev.rhs) { if (int rc = ensure_optional(h, g, OPT_RHS)) return rc; }
  const size_t bytes = (size_t)g->batch * g->plan.n * sizeof(double);
  if (rhs == g->dev.rhs) return 0;
  PP_HIP(hipMemcpyAsync(g->dev.rhs, rhs, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
  return 0;
}

double* pp_rhs_buffer(pp_handle h, int group) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done || alloc_value_storage(h)) return nullptr;
  if (!g->dev.rhs && ensure_optional(h, g, OPT_RHS)) return nullptr;
  return g->dev.rhs;
}

int pp_solve_forward(pp_handle h) {
  if (!h || !h->numeric_done) return fail(h, 3, "pp_solve_forward before numeric factorization");
  PP_HIP(hipSetDevice(h->device));
  hipStream_t st = h->stream;
  const int nc = h->nc;
  // r_s is zeroed only if some group scatters into it (mapped groups: atomic adds) or if there is nothing to store;
  // otherwise the reduction of the first group stores (a memset node costs 5-20 us of stream time around its 1.5 us)
  bool rs_store_first = nc > 0 && !h->groups.empty();
  for (Group* g : h->groups) rs_store_first = rs_store_first && !g->dev.cmapT && g->dev.nc == nc;
  if (!rs_store_first) PP_HIP(hipMemsetAsync(h->rs, 0, std::max<size_t>(nc, 1) * sizeof(double), st));
  GroupStreams gst;
  if (fork_group_streams(h, gst)) return fail(h, 3, "stream fork failed");
  auto group_body = [&](size_t gi) -> int {
    Group* g = h->groups[gi];
    const hipStream_t st = gst.st[gi % (size_t)gst.n];
    const pp::Plan& P = g->plan;
    GroupDev& d = g->dev;
    const bool native = g->rhs_native != nullptr;
    GroupDev dn = d;
    if (native) { dn.rhsN = g->rhs_native; dn.sfwd_zcol = g->zcolN_f; dn.crow_zcol = g->zcolN_c; }
    {
      int nl = native ? 0 : 1;
      for (int l = 0; l < P.n_levels; ++l) nl += (P.clevel_ptr[l + 1] > P.clevel_ptr[l]) && g->fwd_level_has_entries[(size_t)l];
      PhaseScope ps(h, 4, nl);
      if (!native && !d.rhs) return fail(h, 3, "pp_solve_forward: no right-hand side uploaded");
      if (!native) {
        const int tiles = transpose_tiles(P.n, d.nchunk);
        // the right-hand side goes straight to Y in the new (elimination) order: y is then computed in place and
        // the columns without incoming entries (level 0) need no launch at all
        hipLaunchKernelGGL(k_transpose_in, dim3((unsigned)((P.n + 64 * tiles - 1) / (64 * tiles)) * d.nchunk), dim3(256), 0, st, d.rhs,
                           d.Y, d.iperm, d.batch, P.n, d.bpad, tiles, (const int*)nullptr);
      }
      // (a persistent one-workgroup-per-chunk kernel for the small top levels was measured slower than
      // per-level launches: 16 waves on one CU serialise their memory round trips)
      const Splits sp = make_splits(h, d.nchunk);
      hipStream_t fan[PP_MAX_SPLIT];
      if (fork_streams(h, sp, fan, st)) return fail(h, 3, "stream fork failed");
      for (int l = 0; l < P.n_levels; ++l) {
        const int c0 = P.clevel_ptr[l], ncol = P.clevel_ptr[l + 1] - c0;
        if (ncol <= 0 || !g->fwd_level_has_entries[(size_t)l]) continue;
        const int team = g->fwd_level_team[(size_t)l];
        for (int q = 0; q < sp.n; ++q) {
          const int ny = sp.c0[q + 1] - sp.c0[q];
#define PP_LAUNCH_FWD(NW) hipLaunchKernelGGL(k_fwd_level<NW>, dim3((unsigned)ncol * ny), dim3(64 * NW), 0, fan[q], dn, c0, sp.c0[q], ny)
          if (team == 16) PP_LAUNCH_FWD(16);
          else if (team == 4) PP_LAUNCH_FWD(4);
          else if (h->lane_pairs && g->fwd_level_maxrow[(size_t)l] <= PP_PAIR_MAXROW && ny % 2 == 0 && sp.c0[q] % 2 == 0)
            hipLaunchKernelGGL(k_fwd_level_pair, dim3((unsigned)ncol * (ny / 2)), dim3(64), 0, fan[q], dn, c0, sp.c0[q] / 2, ny / 2);
          else PP_LAUNCH_FWD(1);
#undef PP_LAUNCH_FWD
        }
      }
      if (join_streams(h, sp, fan)) return fail(h, 3, "stream join failed");
    }
    return 0;
  };
  if (int rc = run_groups(h, gst, group_body)) return rc;
  if (join_group_streams(h, gst)) return fail(h, 3, "stream join failed");
  // the coupling rows of the groups meet in r_s: one after the other on the handle's stream
  for (Group* g : h->groups) {
    GroupDev& d = g->dev;
    GroupDev dn = d;
    if (g->rhs_native != nullptr) { dn.rhsN = g->rhs_native; dn.sfwd_zcol = g->zcolN_f; dn.crow_zcol = g->zcolN_c; }
    if (d.nc > 0) {
      PhaseScope ps(h, 5, 2);
      hipLaunchKernelGGL(k_fwd_coupling, dim3((unsigned)d.nc * d.nchunk), dim3(64), 0, h->stream, dn, h->rs);
      if (!d.cmapT)
        hipLaunchKernelGGL(k_rs_reduce, dim3((d.nc + 255) / 256), dim3(256), 0, h->stream, d, h->rs,
                           (rs_store_first && g == h->groups.front()) ? 1 : 0);
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
  if (h->btd) {
    PhaseScope psb(h, 6, 1);
    {
      const int gs = h->gs, G = h->G, nlev = (int)h->bcr_ne.size();
      double* b = h->btd_vec + 2 * (size_t)nc + 16;
      double* w = b + nc + 16;
      hipLaunchKernelGGL(k_bcr_rhs, dim3((nc + 255) / 256), dim3(256), 0, st, nc, rc_host ? h->rcd : nullptr, h->rs, b);
      for (int l = 0; l < nlev; ++l) {
        const BcrLevel lv{h->btd_elim + h->bcr_off[(size_t)l], h->bcr_ne[(size_t)l], h->bcr_s[(size_t)l], h->bcr_lo[(size_t)l]};
        for (int phase = 0; phase < (l + 1 < nlev ? 3 : 1); ++phase)
          hipLaunchKernelGGL(k_bcr_fwd, dim3(lv.ne), dim3(512), 0, st, gs, G, lv, phase, h->btd_inv, h->btd_klo, h->btd_kup, b, w);
      }
      for (int l = nlev - 1; l >= 0; --l) {
        const BcrLevel lv{h->btd_elim + h->bcr_off[(size_t)l], h->bcr_ne[(size_t)l], h->bcr_s[(size_t)l], h->bcr_lo[(size_t)l]};
        hipLaunchKernelGGL(k_bcr_bwd, dim3(lv.ne), dim3(512), 0, st, gs, G, lv, h->btd_ylo, h->btd_yup, w, h->xc);
      }
    }
    PP_HIP(hipGetLastError());
    return 0;
  }
  PhaseScope ps(h, 6, 1);
  if (nc > BK_THREADS && nc <= 1024)
    hipLaunchKernelGGL((k_coupling_solve<1024, 16>), dim3(1), dim3(1024), (size_t)nc * sizeof(double), st, nc, h->Sfac,
                       h->ipiv, h->Sldl, h->dvec, h->dense_mode, rc_host ? h->rcd : nullptr, h->rs, h->xc);
  else
    hipLaunchKernelGGL((k_coupling_solve<BK_THREADS, 32>), dim3(1), dim3(BK_THREADS), (size_t)nc * sizeof(double), st, nc,
                       h->Sfac, h->ipiv, h->Sldl, h->dvec, h->dense_mode, rc_host ? h->rcd : nullptr, h->rs, h->xc);
  PP_HIP(hipGetLastError());
  return 0;
}

int pp_solve_coupling_dev(pp_handle h, const double* rc_dev) {
  if (!h || !h->schur_done) return fail(h, 3, "pp_solve_coupling_dev before pp_factor_schur");
  PP_HIP(hipSetDevice(h->device));
  hipStream_t st = h->stream;
  const int nc = h->nc;
  if (nc == 0) return 0;
  if (h->btd) {
    PhaseScope psb(h, 6, 1);
    {
      const int gs = h->gs, G = h->G, nlev = (int)h->bcr_ne.size();
      double* b = h->btd_vec + 2 * (size_t)nc + 16;
      double* w = b + nc + 16;
      hipLaunchKernelGGL(k_bcr_rhs, dim3((nc + 255) / 256), dim3(256), 0, st, nc, rc_dev, h->rs, b);
      for (int l = 0; l < nlev; ++l) {
        const BcrLevel lv{h->btd_elim + h->bcr_off[(size_t)l], h->bcr_ne[(size_t)l], h->bcr_s[(size_t)l], h->bcr_lo[(size_t)l]};
        for (int phase = 0; phase < (l + 1 < nlev ? 3 : 1); ++phase)
          hipLaunchKernelGGL(k_bcr_fwd, dim3(lv.ne), dim3(512), 0, st, gs, G, lv, phase, h->btd_inv, h->btd_klo, h->btd_kup, b, w);
      }
      for (int l = nlev - 1; l >= 0; --l) {
        const BcrLevel lv{h->btd_elim + h->bcr_off[(size_t)l], h->bcr_ne[(size_t)l], h->bcr_s[(size_t)l], h->bcr_lo[(size_t)l]};
        hipLaunchKernelGGL(k_bcr_bwd, dim3(lv.ne), dim3(512), 0, st, gs, G, lv, h->btd_ylo, h->btd_yup, w, h->xc);
      }
    }
    PP_HIP(hipGetLastError());
    return 0;
  }
  PhaseScope ps(h, 6, 1);
  if (nc > BK_THREADS && nc <= 1024)
    hipLaunchKernelGGL((k_coupling_solve<1024, 16>), dim3(1), dim3(1024), (size_t)nc * sizeof(double), st, nc, h->Sfac,
                       h->ipiv, h->Sldl, h->dvec, h->dense_mode, rc_dev, h->rs, h->xc);
  else
    hipLaunchKernelGGL((k_coupling_solve<BK_THREADS, 32>), dim3(1), dim3(BK_THREADS), (size_t)nc * sizeof(double), st, nc,
                       h->Sfac, h->ipiv, h->Sldl, h->dvec, h->dense_mode, rc_dev, h->rs, h->xc);
  PP_HIP(hipGetLastError());
  return 0;
}

double* pp_coupling_solution_buffer(pp_handle h) { return (h && h->symbolic_done) ? h->xc : nullptr; }

int pp_copy_coupling_solution(pp_handle h, double* dev_ptr) {
  if (!h || !h->schur_done) return fail(h, 3, "pp_copy_coupling_solution before pp_factor_schur");
  PP_HIP(hipSetDevice(h->device));
  if (h->nc > 0) PP_HIP(hipMemcpyAsync(dev_ptr, h->xc, (size_t)h->nc * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  return 0;
}

int pp_bind_solution_buffer(pp_handle h, int group, double* dev_ptr) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_bind_solution_buffer: bad group");
  if (int rc = alloc_value_storage(h)) return rc;
  g->dev.xout = dev_ptr ? dev_ptr : g->xout_own;
  return 0;
}

int pp_bind_native_vectors(pp_handle h, int group, const double* rhs_dev, double* x_dev) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_bind_native_vectors: bad group");
  if ((rhs_dev == nullptr) != (x_dev == nullptr)) return fail(h, 3, "pp_bind_native_vectors: give both buffers or neither");
  if (int rc = alloc_value_storage(h)) return rc;
  g->rhs_native = rhs_dev;
  g->x_native = x_dev;
  return 0;
}

int pp_solve_backward(pp_handle h) {
  if (!h || !h->schur_done) return fail(h, 3, "pp_solve_backward before pp_factor_schur");
  PP_HIP(hipSetDevice(h->device));
  GroupStreams gst;
  if (fork_group_streams(h, gst)) return fail(h, 3, "stream fork failed");
  auto group_body = [&](size_t gi) -> int {
    Group* g = h->groups[gi];
    const hipStream_t st = gst.st[gi % (size_t)gst.n];
    const pp::Plan& P = g->plan;
    GroupDev& d = g->dev;
    const bool native = g->x_native != nullptr;
    if (!native && (!d.xout || !d.X)) { if (int rc = ensure_optional(h, g, (d.xout ? 0 : OPT_XOUT) | OPT_X)) return rc; }
    int nlb = native ? 0 : 1;
    for (int l = 0; l < P.n_levels; ++l) nlb += P.clevel_ptr[l + 1] > P.clevel_ptr[l];
    PhaseScope ps(h, 7, nlb + ((d.cmapT && d.nc > 0) ? 1 : 0));
    GroupDev dn = d;
    if (native) { dn.rhsN = g->rhs_native; dn.bwd_rec = g->brecN; dn.rowidx = g->rowidx_o; dn.X = g->x_native; }
    const double* xcp = h->xc;
    if (d.cmapT && d.nc > 0) {     // mapped group: every instance reads the coupling values of its own rows
      hipLaunchKernelGGL(k_gather_xc, dim3((unsigned)(((size_t)d.nc * d.bpad + 255) / 256)), dim3(256), 0, st, d, h->xc);
      xcp = d.XCL;
    }
    {
      const Splits sp = make_splits(h, d.nchunk);
      hipStream_t fan[PP_MAX_SPLIT];
      if (fork_streams(h, sp, fan, st)) return fail(h, 3, "stream fork failed");
      for (int l = P.n_levels - 1; l >= 0; --l) {
        const int c0 = P.clevel_ptr[l], ncol = P.clevel_ptr[l + 1] - c0;
        if (ncol <= 0) continue;
        const int team = g->bwd_level_team[(size_t)l];
        for (int q = 0; q < sp.n; ++q) {
          const int ny = sp.c0[q + 1] - sp.c0[q];
#define PP_LAUNCH_BWD(NW) hipLaunchKernelGGL(k_bwd_level<NW>, dim3((unsigned)ncol * ny), dim3(64 * NW), 0, fan[q], dn, c0, sp.c0[q], ny, xcp)
          if (team == 16) PP_LAUNCH_BWD(16);
          else if (team == 4) PP_LAUNCH_BWD(4);
          else if (h->lane_pairs && g->bwd_level_maxrow[(size_t)l] <= PP_PAIR_MAXROW && ny % 2 == 0 && sp.c0[q] % 2 == 0)
            hipLaunchKernelGGL(k_bwd_level_pair, dim3((unsigned)ncol * (ny / 2)), dim3(64), 0, fan[q], dn, c0, sp.c0[q] / 2, ny / 2, xcp);
          else PP_LAUNCH_BWD(1);
#undef PP_LAUNCH_BWD
        }
      }
      if (join_streams(h, sp, fan)) return fail(h, 3, "stream join failed");
    }
    if (!native)
      hipLaunchKernelGGL(k_transpose_out, dim3((unsigned)((P.n + 63) / 64) * d.nchunk), dim3(256), 0, st, d.X, d.iperm, d.xout,
                         d.batch, P.n, d.bpad);
    return 0;
  };
  if (int rc = run_groups(h, gst, group_body)) return rc;
  if (join_group_streams(h, gst)) return fail(h, 3, "stream join failed");
  PP_HIP(hipGetLastError());
  return 0;
}

int pp_download_solution(pp_handle h, int group, double* x, int on_device) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_download_solution: bad group");
  PP_HIP(hipSetDevice(h->device));
  const size_t bytes = (size_t)g->batch * g->plan.n * sizeof(double);
  if (!g->dev.xout) return fail(h, 3, "pp_download_solution: no solution in the [instance][row] layout (native vectors bound?)");
  if (x != g->dev.xout)
    PP_HIP(hipMemcpyAsync(x, g->dev.xout, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
  if (!on_device) PP_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

double* pp_solution_buffer(pp_handle h, int group) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done || alloc_value_storage(h)) return nullptr;
  if (!g->dev.xout && ensure_optional(h, g, OPT_XOUT)) return nullptr;
  return g->dev.xout;
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
  g->input_mode = Group::IN_RAW;
  return 0;
}

int pp_bind_rhs_buffer(pp_handle h, int group, double* dev_ptr) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_bind_rhs_buffer: bad group");
  g->dev.rhs = dev_ptr ? dev_ptr : g->rhs_own;
  return 0;
}

int pp_set_supernodes(pp_handle h, int wmax, int tol_rows) {
  if (!h) return 3;
  if (wmax < 0 || wmax > PP_WMAX) return fail(h, 3, "supernode width must be 0 (default) .. PP_WMAX");
  h->sn_wmax = wmax;
  h->sn_tol = tol_rows;
  return 0;
}

int pp_set_instance_splits(pp_handle h, int nsplit) {
  if (!h) return 3;
  if (nsplit < 0 || nsplit > PP_MAX_SPLIT) return fail(h, 3, "instance splits must be 0 (automatic) .. 8");
  h->nsplit_req = nsplit;
  return 0;
}

int pp_set_dense_policy(pp_handle h, int policy) {
  if (!h) return 3;
  if (policy != 0 && policy != 1) return fail(h, 3, "dense policy must be 0 (auto) or 1 (Bunch-Kaufman only)");
  h->dense_policy = policy;
  return 0;
}

int pp_get_dense_mode(pp_handle h, int* mode_out) {
  if (!h || !h->schur_done) return fail(h, 3, "pp_get_dense_mode before pp_factor_schur");
  PP_HIP(hipSetDevice(h->device));
  PP_HIP(hipMemcpyAsync(mode_out, h->dense_mode, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  PP_HIP(hipStreamSynchronize(h->stream));
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
  if (!(factor > 0.0)) return fail(h, 3, "memory allocation factor must be positive");
  h->mem_factor *= factor;
  return 0;
}

int pp_set_memory_budget(pp_handle h, int64_t bytes) {
  if (!h) return 3;
  if (bytes < 0) return fail(h, 3, "memory budget must be >= 0 (0: no limit)");
  h->mem_budget = bytes;
  h->mem_factor = 1.0;
  return 0;
}

int pp_memory_info(pp_handle h, int64_t out[3]) {
  if (!h) return 3;
  out[0] = h->mem_required;
  out[1] = h->mem_budget > 0 ? (int64_t)((double)h->mem_budget * h->mem_factor) : 0;
  int64_t allocated = 0;      // what is allocated now: the optional input / output copies only once something used them
  if (h->values_allocated) {
    allocated = h->mem_required;
    for (Group* g : h->groups) {
      const int64_t bp = g->dev.bpad;
      if (!g->raw_own) allocated -= 8 * (int64_t)g->batch * g->nraw;
      if (!g->rawT_own) allocated -= 8 * (int64_t)std::max(g->nraw_used, 1) * bp;
      if (!g->rhs_own) allocated -= 8 * (int64_t)g->batch * g->plan.n;
      if (!g->xout_own) allocated -= 8 * (int64_t)g->batch * g->plan.n;
      if (!g->dev.X) allocated -= 8 * (int64_t)g->plan.n * bp;
    }
  }
  out[2] = allocated;
  return 0;
}

int pp_bcr_block_paths(pp_handle h, int32_t out[2]) {
  if (!h) return 3;
  out[0] = out[1] = 0;
  if (!h->btd || !h->schur_done || !h->btd_info) return 0;
  PP_HIP(hipSetDevice(h->device));
  std::vector<int> info(4 * (size_t)h->G);
  PP_HIP(hipMemcpyAsync(info.data(), h->btd_info, info.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  PP_HIP(hipStreamSynchronize(h->stream));
  for (int t = 0; t < h->G; ++t) out[info[4 * (size_t)t + 3] == 1 ? 0 : 1] += 1;
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
                         P.flops_schur, (int64_t)P.ftasks.size(), (int64_t)P.fentries.size(), (int64_t)P.stile_a.size(),
                         (int64_t)P.stile_rec.size(), P.ncan, g->nraw};
  std::memcpy(out, v, sizeof(v));
  return 0;
}


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

#ifdef PP_X_STAMPS
int pp_x_set_stamps(pp_handle h, void* dev_buffer, int level) {
  const pp::Plan& P = h->groups[0]->plan;
  const int task0 = (level >= 0 && level < P.n_levels) ? P.flevel_ptr[level] : -1;
  if (task0 < 0) {
    unsigned long long* ptr0 = (unsigned long long*)dev_buffer;
    PP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(pp_x_stamps), &ptr0, sizeof(ptr0)));
    PP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(pp_x_stamp_task0), &task0, sizeof(task0)));
    return 0;
  }
  unsigned long long* ptr = (unsigned long long*)dev_buffer;
  PP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(pp_x_stamps), &ptr, sizeof(ptr)));
  PP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(pp_x_stamp_task0), &task0, sizeof(task0)));
  return P.flevel_ptr[level + 1] - P.flevel_ptr[level];
}
#endif

int pp_group_stats_ex(pp_handle h, int group, int64_t out[16]) {
  Group* g = get_group(h, group);
  if (!g) return fail(h, 3, "pp_group_stats_ex: bad group");
  const pp::Plan& P = g->plan;
  int64_t coupling_entries = 0;
  for (int p = 0; p < P.npiv; ++p) coupling_entries += (int64_t)P.piv_ncrow[p] * P.piv_w[p];
  int64_t launches_factor = 0, launches_fwd = 1, launches_bwd = 1;
  for (int l = 0; l < P.n_levels; ++l) {
    launches_factor += (P.flevel_ptr[l + 1] > P.flevel_ptr[l]) + (P.slevel_ptr[l + 1] > P.slevel_ptr[l]) +
                       ((P.front_piv >= 0 && P.piv_level[P.front_piv] == l) ? 1 + (P.wtasks.empty() ? 0 : 1) : 0);
    if (l < (int)g->fwd_level_has_entries.size() && g->fwd_level_has_entries[(size_t)l]) ++launches_fwd;
    if (P.clevel_ptr[l + 1] > P.clevel_ptr[l]) ++launches_bwd;
  }
  const int64_t index_bytes = 4 * ((int64_t)P.fentries.size() * 4 + (int64_t)P.ftasks.size() * TASK_INTS +
                                   (int64_t)P.stasks.size() * TASK_INTS + (int64_t)P.fdst_ptr.size() +
                                   2 * (int64_t)P.sfwd_upos.size() + 2 * (int64_t)P.crow_upos.size() + (int64_t)P.rowidx.size() +
                                   12 * (int64_t)P.n + 20 * (int64_t)P.stile_rec.size() * 4);
  const int64_t v[16] = {g->nraw_used, P.dsize, P.bsize, coupling_entries, index_bytes, (int64_t)P.sfwd_upos.size(),
                         (int64_t)P.crow_upos.size(), g->nsrc, launches_factor, launches_fwd, launches_bwd,
                         (int64_t)g->dev.bpad, (int64_t)g->dev.nchunk, (int64_t)g->ntiles, (int64_t)P.tail_level0, 0};
  std::memcpy(out, v, sizeof(v));
  return 0;
}

int pp_group_perm(pp_handle h, int group, int32_t* perm) {
  Group* g = get_group(h, group);
  if (!g) return fail(h, 3, "pp_group_perm: bad group");
  std::memcpy(perm, g->plan.perm.data(), sizeof(int) * g->plan.n);
  return 0;
}

int pp_set_diagonal_classes(pp_handle h, int group, const int8_t* cls) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_set_diagonal_classes: bad group or symbolic phase not finished");
  PP_HIP(hipSetDevice(h->device));
  std::vector<int> rows, kinds;
  for (int i = 0; i < g->plan.n; ++i) {
    if (cls[i] == 0) continue;
    if (cls[i] != 1 && cls[i] != 2) return fail(h, 3, "pp_set_diagonal_classes: classes are 0, 1 (Hessian) or 2 (constraint)");
    const int ce = g->diag_can[(size_t)i];
    if (ce < 0)
      return fail(h, 3, "pp_set_diagonal_classes: row " + std::to_string(i) +
                            " has a class but no diagonal entry in the planned pattern");
    // the shift goes to the first raw duplicate of the canonical diagonal entry
    const int raw = g->can_idx[(size_t)g->can_ptr[(size_t)ce]];
    rows.push_back(-1 - raw);
    kinds.push_back((int)cls[i]);
  }
  // raw index -> compact row of the transposed input (same rule as pp_end_symbolic)
  {
    std::vector<int> rawmap((size_t)std::max(g->nraw, 1), -1);
    for (size_t j = 0; j < g->used_raw.size(); ++j) rawmap[(size_t)g->used_raw[j]] = (int)j;
    for (auto& r : rows) r = rawmap[(size_t)(-1 - r)];
  }
  PP_HIP(hipStreamSynchronize(h->stream));      // a previous shifted factorisation may still read the old arrays
  if (g->shift_row) { (void)hipFree(g->shift_row); g->shift_row = nullptr; }
  if (g->shift_cls) { (void)hipFree(g->shift_cls); g->shift_cls = nullptr; }
  g->nshift = 0;
  rows.push_back(0); kinds.push_back(0);
  int rc;
  if ((rc = dev_alloc(h, (Group*)nullptr, &g->shift_row, rows.size()))) return rc;
  if ((rc = dev_alloc(h, (Group*)nullptr, &g->shift_cls, kinds.size()))) return rc;
  PP_HIP(hipMemcpy(g->shift_row, rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice));
  PP_HIP(hipMemcpy(g->shift_cls, kinds.data(), kinds.size() * sizeof(int), hipMemcpyHostToDevice));
  g->nshift = (int)rows.size() - 1;
  return 0;
}

int pp_numeric_local_shifted(pp_handle h, double delta_w, double delta_c) {
  if (!h) return 3;
  h->shift_w = delta_w;
  h->shift_c = delta_c;
  const int rc = pp_numeric_local(h);
  h->shift_w = 0.0;
  h->shift_c = 0.0;
  return rc;
}

// Host-side staging (no device work): for every block whose raw COO index arrays equal the group's reference
// arrays, the values go to the block's row of the staging array; same_out[i] tells the caller which blocks it has to
// canonicalise itself (quirk Q7).  Compare + copy are memory-bound, so they are spread over host threads.
// runs (may be null = everything): triples {first entry, length, destination offset in the row} of the K data and of
// the border data that are copied -- the entries some canonical entry reads (a KKT block handed over with both
// triangles has whole runs of upper-triangle entries nobody reads: they are neither staged nor uploaded).
namespace {
struct StageArgs {
  const int32_t* const* kr; const int32_t* const* kc; const double* const* kd; const int64_t* knnz;
  const int32_t* const* br; const int32_t* const* bc; const double* const* bd; const int64_t* bnnz;
  const int32_t *ref_kr, *ref_kc; int64_t ref_knnz; const int32_t *ref_br, *ref_bc; int64_t ref_bnnz;
  int nrunsK; const int64_t* runsK; int nrunsB; const int64_t* runsB;
  double* staging; int64_t row_stride; const int32_t* slots; uint8_t* same_out;
};

void stage_range(const StageArgs& a, int i0, int i1) {
  for (int i = i0; i < i1; ++i) {
    bool same = a.knnz[i] == a.ref_knnz && a.bnnz[i] == a.ref_bnnz;
    const size_t kb = (size_t)a.ref_knnz * sizeof(int32_t), bb = (size_t)a.ref_bnnz * sizeof(int32_t);
    same = same && (a.kr[i] == a.ref_kr || kb == 0 || std::memcmp(a.kr[i], a.ref_kr, kb) == 0);
    same = same && (a.kc[i] == a.ref_kc || kb == 0 || std::memcmp(a.kc[i], a.ref_kc, kb) == 0);
    same = same && (a.br[i] == a.ref_br || bb == 0 || std::memcmp(a.br[i], a.ref_br, bb) == 0);
    same = same && (a.bc[i] == a.ref_bc || bb == 0 || std::memcmp(a.bc[i], a.ref_bc, bb) == 0);
    if (same) {
      double* row = a.staging + (size_t)a.slots[i] * (size_t)a.row_stride;
      if (!a.runsK) {
        if (a.ref_knnz > 0) std::memcpy(row, a.kd[i], (size_t)a.ref_knnz * sizeof(double));
        if (a.ref_bnnz > 0) std::memcpy(row + a.ref_knnz, a.bd[i], (size_t)a.ref_bnnz * sizeof(double));
      } else {
        for (int r = 0; r < a.nrunsK; ++r)
          std::memcpy(row + a.runsK[3 * r + 2], a.kd[i] + a.runsK[3 * r], (size_t)a.runsK[3 * r + 1] * sizeof(double));
        for (int r = 0; r < a.nrunsB; ++r)
          std::memcpy(row + a.runsB[3 * r + 2], a.bd[i] + a.runsB[3 * r], (size_t)a.runsB[3 * r + 1] * sizeof(double));
      }
    }
    a.same_out[i] = same ? 1 : 0;
  }
}

void stage_parallel(const StageArgs& a, int i0, int i1, int nthreads) {
  const int n = i1 - i0;
  const int nt = std::max(1, std::min(std::min(nthreads, 64), n));
  if (nt == 1) { stage_range(a, i0, i1); return; }
  std::vector<std::thread> pool;
  pool.reserve((size_t)nt);
  int started = 0;
  try {                           // (no exception may cross the C ABI: what could not be started runs here)
    for (; started < nt; ++started)
      pool.emplace_back(stage_range, std::cref(a), i0 + (int)((int64_t)n * started / nt), i0 + (int)((int64_t)n * (started + 1) / nt));
  } catch (...) {
  }
  if (started < nt) stage_range(a, i0 + (int)((int64_t)n * started / nt), i1);
  for (auto& th : pool) th.join();
}

bool stage_args_ok(int nblocks, const StageArgs& a, int64_t need) {
  if (nblocks < 0 || !a.same_out) return false;
  if (nblocks > 0 && (!a.kr || !a.kc || !a.kd || !a.knnz || !a.br || !a.bc || !a.bd || !a.bnnz || !a.staging || !a.slots)) return false;
  if ((a.nrunsK > 0 && !a.runsK) || (a.nrunsB > 0 && !a.runsB)) return false;
  for (int r = 0; r < a.nrunsK; ++r)
    if (a.runsK[3 * r] < 0 || a.runsK[3 * r + 1] < 0 || a.runsK[3 * r] + a.runsK[3 * r + 1] > a.ref_knnz || a.runsK[3 * r + 2] < 0 ||
        a.runsK[3 * r + 2] + a.runsK[3 * r + 1] > a.row_stride) return false;
  for (int r = 0; r < a.nrunsB; ++r)
    if (a.runsB[3 * r] < 0 || a.runsB[3 * r + 1] < 0 || a.runsB[3 * r] + a.runsB[3 * r + 1] > a.ref_bnnz || a.runsB[3 * r + 2] < 0 ||
        a.runsB[3 * r + 2] + a.runsB[3 * r + 1] > a.row_stride) return false;
  return need <= a.row_stride;
}
}  // namespace

int pp_stage_values(int nblocks, int nthreads, const int32_t* const* kr, const int32_t* const* kc,
                    const double* const* kd, const int64_t* knnz, const int32_t* const* br, const int32_t* const* bc,
                    const double* const* bd, const int64_t* bnnz, const int32_t* ref_kr, const int32_t* ref_kc,
                    int64_t ref_knnz, const int32_t* ref_br, const int32_t* ref_bc, int64_t ref_bnnz, double* staging,
                    int64_t row_stride, const int32_t* slots, uint8_t* same_out) {
  const StageArgs a{kr, kc, kd, knnz, br, bc, bd, bnnz, ref_kr, ref_kc, ref_knnz, ref_br, ref_bc, ref_bnnz, 0, nullptr, 0,
                    nullptr, staging, row_stride, slots, same_out};
  if (!stage_args_ok(nblocks, a, ref_knnz + ref_bnnz)) return 3;
  stage_parallel(a, 0, nblocks, nthreads);
  return 0;
}

int pp_stage_values_runs(int nblocks, int nthreads, const int32_t* const* kr, const int32_t* const* kc,
                         const double* const* kd, const int64_t* knnz, const int32_t* const* br, const int32_t* const* bc,
                         const double* const* bd, const int64_t* bnnz, const int32_t* ref_kr, const int32_t* ref_kc,
                         int64_t ref_knnz, const int32_t* ref_br, const int32_t* ref_bc, int64_t ref_bnnz, int nruns_k,
                         const int64_t* runs_k, int nruns_b, const int64_t* runs_b, double* staging, int64_t row_stride,
                         const int32_t* slots, uint8_t* same_out) {
  const StageArgs a{kr, kc, kd, knnz, br, bc, bd, bnnz, ref_kr, ref_kc, ref_knnz, ref_br, ref_bc, ref_bnnz, nruns_k, runs_k,
                    nruns_b, runs_b, staging, row_stride, slots, same_out};
  if (!runs_k || !stage_args_ok(nblocks, a, 0)) return 3;
  stage_parallel(a, 0, nblocks, nthreads);
  return 0;
}

// The same with the upload overlapped: the blocks (ascending slots) are staged in slices and every finished slice of
// rows goes to the device with an asynchronous copy while the host threads stage the next one (the staging array
// must be pinned for the copies to be asynchronous).  The rows of blocks reported in same_out as not staged are
// uploaded by the caller afterwards (pp_upload_values_compact on their row range).
int pp_stage_upload_compact(pp_handle h, int group, int nblocks, int nthreads, const int32_t* const* kr,
                            const int32_t* const* kc, const double* const* kd, const int64_t* knnz,
                            const int32_t* const* br, const int32_t* const* bc, const double* const* bd,
                            const int64_t* bnnz, const int32_t* ref_kr, const int32_t* ref_kc, int64_t ref_knnz,
                            const int32_t* ref_br, const int32_t* ref_bc, int64_t ref_bnnz, int nruns_k, const int64_t* runs_k,
                            int nruns_b, const int64_t* runs_b, double* staging, const int32_t* slots, uint8_t* same_out) {
  Group* g = get_group(h, group);
  if (!g || !h->symbolic_done) return fail(h, 3, "pp_stage_upload_compact: bad group or symbolic phase not finished");
  const StageArgs a{kr, kc, kd, knnz, br, bc, bd, bnnz, ref_kr, ref_kc, ref_knnz, ref_br, ref_bc, ref_bnnz, nruns_k, runs_k,
                    nruns_b, runs_b, staging, (int64_t)g->nraw_used, slots, same_out};
  if (!runs_k || !stage_args_ok(nblocks, a, 0)) return fail(h, 3, "pp_stage_upload_compact: bad arguments");
  for (int i = 0; i < nblocks; ++i)
    if (slots[i] < 0 || slots[i] >= g->batch || (i > 0 && slots[i] <= slots[i - 1]))
      return fail(h, 3, "pp_stage_upload_compact: slots must be ascending and inside the batch");
  PP_HIP(hipSetDevice(h->device));
  if (int rc = ensure_optional(h, g, OPT_RAW)) return rc;
  g->input_mode = Group::IN_COMPACT;
  const int slice = 128;
  for (int i0 = 0; i0 < nblocks; i0 += slice) {
    const int i1 = std::min(nblocks, i0 + slice);
    stage_parallel(a, i0, i1, nthreads);
    const size_t stride = (size_t)g->nraw_used;
    const int r0 = slots[i0], r1 = slots[i1 - 1] + 1;
    if (stride > 0)
      PP_HIP(hipMemcpyAsync(g->raw_own + (size_t)r0 * stride, staging + (size_t)r0 * stride, (size_t)(r1 - r0) * stride * sizeof(double),
                            hipMemcpyHostToDevice, h->stream));
  }
  return 0;
}

// dst[idx[i]][0 .. row_doubles) = src[i][0 .. row_doubles): the right-hand sides of the local blocks into their staging
// rows, on host threads (75 MB per back-solve at the headline size)
int pp_copy_rows(int nrows, int nthreads, const double* const* src, const int64_t* idx, double* dst, int64_t row_doubles) {
  if (nrows < 0 || row_doubles < 0 || (nrows > 0 && (!src || !idx || !dst))) return 3;
  auto work = [&](int i0, int i1) {
    for (int i = i0; i < i1; ++i) std::memcpy(dst + (size_t)idx[i] * (size_t)row_doubles, src[i], (size_t)row_doubles * sizeof(double));
  };
  const int nt = std::max(1, std::min(std::min(nthreads, 64), nrows));
  if (nt == 1) { work(0, nrows); return 0; }
  std::vector<std::thread> pool;
  pool.reserve((size_t)nt);
  int started = 0;
  try {
    for (; started < nt; ++started)
      pool.emplace_back(work, (int)((int64_t)nrows * started / nt), (int)((int64_t)nrows * (started + 1) / nt));
  } catch (...) {
  }
  if (started < nt) work((int)((int64_t)nrows * started / nt), nrows);
  for (auto& th : pool) th.join();
  return 0;
}

// pinned host memory for staging arrays / result buffers of the host boundary (hipHostMalloc; NULL on failure)
void* pp_host_alloc(int64_t bytes) {
  void* p = nullptr;
  if (bytes <= 0 || hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}
void pp_host_free(void* p) { if (p) (void)hipHostFree(p); }

int pp_set_pivot_tolerance(pp_handle h, double u_symbolic, double u_runtime) {
  if (!h) return 3;
  if (u_symbolic < 0.0 || u_symbolic > 0.5 || u_runtime < 0.0 || u_runtime > 0.5)
    return fail(h, 3, "pivot tolerances must lie in [0, 0.5] (0: default / off)");
  h->pivot_threshold = u_symbolic;
  h->growth_bound = u_runtime > 0.0 ? 1.0 / u_runtime : 1e8;
  h->growth_fatal = u_runtime > 0.0;
  return 0;
}

int pp_get_growth_count(pp_handle h, int64_t* out) {
  if (!h || !h->schur_done || !out) return fail(h, 3, "pp_get_growth_count before pp_factor_schur");
  *out = (int64_t)h->status_host[5];     // (valid once pp_get_status has seen the mailbox of this factorisation)
  return 0;
}

int pp_find_growth(pp_handle h, int group, int32_t* instance_out) {
  Group* g = get_group(h, group);
  if (!g || !instance_out || !h->numeric_done) return fail(h, 3, "pp_find_growth: bad group or no numeric factorization");
  const GroupDev& d = g->dev;
  *instance_out = -1;
  PP_HIP(hipSetDevice(h->device));
  PP_HIP(hipStreamSynchronize(h->stream));
  std::vector<int> flags((size_t)d.bpad);
  PP_HIP(hipMemcpy(flags.data(), d.growth + d.bpad, flags.size() * sizeof(int), hipMemcpyDeviceToHost));
  for (int b = 0; b < d.batch; ++b)
    if (flags[(size_t)b]) { *instance_out = b; break; }
  return 0;
}

int pp_find_zero_pivot(pp_handle h, int group, int32_t* instance_out) {
  Group* g = get_group(h, group);
  if (!g || !instance_out || !h->numeric_done) return fail(h, 3, "pp_find_zero_pivot: bad group or no numeric factorization");
  const GroupDev& d = g->dev;
  *instance_out = -1;
  PP_HIP(hipSetDevice(h->device));
  PP_HIP(hipStreamSynchronize(h->stream));
  // rare path (a factorisation that reported numerically zero pivots): the 16-bit codes come to the host as they are
  std::vector<unsigned short> codes((size_t)g->plan.npiv * d.bpad);
  PP_HIP(hipMemcpy(codes.data(), d.codes, codes.size() * sizeof(unsigned short), hipMemcpyDeviceToHost));
  for (int p = 0; p < g->plan.npiv && *instance_out < 0; ++p)        // first pivot in elimination order that broke
    for (int b = 0; b < d.batch; ++b)
      if ((codes[(size_t)p * d.bpad + b] >> 8) & 15u) { *instance_out = b; break; }
  return 0;
}

int pp_get_factor(pp_handle h, int group, int which, int instance, double* out, int64_t count) {
  Group* g = get_group(h, group);
  if (!g) return fail(h, 3, "pp_get_factor: bad group");
  const GroupDev& d = g->dev;
  const double* src = which == 0 ? d.U : which == 1 ? d.L : which == 2 ? d.Dinv : which == 3 ? d.rawT : nullptr;
  if (!src) return fail(h, 3, "pp_get_factor: that array does not exist (fused sources: no transposed input)");
  const int64_t rows = which == 2 ? g->plan.dsize : which == 3 ? g->nraw_used : g->plan.usize;
  if (!src || instance < 0 || instance >= d.batch || count > rows) return fail(h, 3, "pp_get_factor: bad arguments");
  PP_HIP(hipSetDevice(h->device));
  PP_HIP(hipStreamSynchronize(h->stream));
  PP_HIP(hipMemcpy2D(out, sizeof(double), src + instance, sizeof(double) * (size_t)d.bpad, sizeof(double), (size_t)count,
                     hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"
