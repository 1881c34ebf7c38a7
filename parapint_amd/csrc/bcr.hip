// Block-tridiagonal S of time-staged problems: block cyclic reduction on the fp64 matrix cores and its solve
// (the reference factorises a sparse COO S with its sub-solver, mpi_explicit_schur_complement.py:88-125, 228-255, 352-361).
#include "common.hpp"
#include "dense_blocks.hpp"

namespace {

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
// pair rotation threshold (k_bcr_ldl_inverse): rotate where |entry between the pair| > theta * max |its diagonals|.
// MEASURED on the blocks of three interior-point runs at every level of the cyclic reduction (largest multiplier of the
// ordered unpivoted factorisation): theta 0.5 -- 1e11 on the reduced blocks of a 96-block quadratic program where only
// 22..26 of 30 pairs rotate; 0.25 / 0.1 / 0.05 -- at most 0.6 / 0.77 / 0.94 on all of them (Burgers: no pair rotates down
// to 0.1, its link duals carry 15..70 on the diagonal; at 0.05 one does, harmlessly).  A pair with |diagonal| >> |entry|
// must NOT rotate (the second rotated pivot is 2 entry^2 / diagonal).
constexpr double BL_ROT_THETA = 0.1;
// ... and only in a block that holds at least one pair with |entry| > BL_ROT_NEED * max |its diagonals| (a block whose pairs
// all carry diagonals of the size of the entry -- the synthetic C4 blocks -- factorises as it is, and the rotation costs
// 6-19 us of the kernel's 55: 4 x the loads of the block and a pass over its inverse).  Same scan: 0.5 ... 1.4 give the same
// largest multipliers as rotating everywhere (<= 0.94; 1.6 at the first iterate of one run, unrotated); 2.0 misses blocks.
constexpr double BL_ROT_NEED = 0.7;
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
  __shared__ int sflags[2], orig[16 * BL_NT], rotf[8 * BL_NT];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = BL_THREADS / 64;
  const int li = lane & 15, lk = lane >> 4;
  const int i = lv.elim[blockIdx.x];
  const double* Dg = D + (size_t)i * gs * gs;
  const int nt = (gs + 15) / 16, ntt = nt * (nt + 1) / 2;
  // ---- symmetric pre-ordering by decreasing |diagonal| (static, from the values of this factorisation: the coupling
  // states of a time-staged S carry O(1) diagonals, the link duals nearly none -- states first makes every pivot of
  // the quasi-definite block its column's largest entry; the acceptance test below judges the result)
  // ---- pair rotations (in front of the ordering): rows k and k + h, h = gs / 2, are a link dual and the coupling state it
  // multiplies; where the entry between them outweighs their diagonals (a quadratic program whose dynamics leave both the
  // dual block and the state block of S nearly empty: [[-p, -1], [-1, q]] with p, q << 1) no ordering of 1 x 1 pivots is
  // stable, but the orthogonal congruence with R = [[1, 1], [1, -1]] / sqrt(2) on that pair turns it into
  // [[(q - p) / 2 - 1, .], [., (q - p) / 2 + 1]]: D' = H D H with H = H^T = H^-1 the direct sum of R (rotated pairs) and 1,
  // inv(D) = H inv(D') H, same inertia.  A pair whose diagonals carry it (the nonlinear Burgers blocks: p = 15..70) is left
  // alone (BL_ROT_THETA).  MEASURED on the blocks of real interior-point runs: largest multiplier
  // 1e2..inf -> < 1.
  const int h2 = gs / 2;
  if (tid == 0) { sflags[0] = 0; sflags[1] = 0; }
  __syncthreads();
  for (int k = tid; k < 16 * BL_NT; k += BL_THREADS) {
    dmag[k] = (k < gs) ? fabs(Dg[(size_t)k + (size_t)k * gs]) : -1.0;
    orig[k] = k;          // (a NaN diagonal leaves ranks unassigned: the identity keeps every index valid, the pivot test rejects)
    if (k < h2) {
      const double a = Dg[(size_t)k + (size_t)k * gs], c = Dg[(size_t)(k + h2) + (size_t)(k + h2) * gs];
      const double b = Dg[(size_t)(k + h2) + (size_t)k * gs];
      rotf[k] = 0;
      if (fabs(b) > BL_ROT_NEED * fmax(fabs(a), fabs(c))) sflags[1] = 1;      // this block needs its pairs rotated
    }
  }
  __syncthreads();
  const bool anyrot = sflags[1] != 0;
  if (anyrot) {
    for (int k = tid; k < h2; k += BL_THREADS) {
      const double a = Dg[(size_t)k + (size_t)k * gs], c = Dg[(size_t)(k + h2) + (size_t)(k + h2) * gs];
      const double b = Dg[(size_t)(k + h2) + (size_t)k * gs];
      if (fabs(b) > BL_ROT_THETA * fmax(fabs(a), fabs(c))) {
        rotf[k] = 1;
        dmag[k] = fabs(0.5 * (a + c) + b);
        dmag[k + h2] = fabs(0.5 * (a + c) - b);
      }
    }
    __syncthreads();
  }
  // entry (r, c) of H D H (r, c: labels of D); hcoef: the coefficient of the label itself, its partner's is 1 / sqrt(2)
  auto rotated = [&](int r) { return r < 2 * h2 && rotf[r < h2 ? r : r - h2] != 0; };
  auto entry = [&](int r, int c) -> double {
    const bool rr = anyrot && rotated(r), rc = anyrot && rotated(c);
    auto D1 = [&](int x, int y) { return Dg[(size_t)max(x, y) + (size_t)min(x, y) * gs]; };
    if (!rr && !rc) return D1(r, c);
    const double sq = 0.70710678118654752440;
    const int pr = r < h2 ? r + h2 : r - h2, pc = c < h2 ? c + h2 : c - h2;
    const double hr = rr ? (r < h2 ? sq : -sq) : 1.0, hc = rc ? (c < h2 ? sq : -sq) : 1.0;
    double v = hr * hc * D1(r, c);
    if (rr) v += sq * hc * D1(pr, c);
    if (rc) v += hr * sq * D1(r, pc);
    if (rr && rc) v += 0.5 * D1(pr, pc);
    return v;
  };
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
    if (gr < gs && gc < gs) v = entry(orig[gr], orig[gc]);
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
  if (anyrot) {
    // inv(D) = H inv(D') H, in place on the 2 x 2 cells (pair I) x (pair J) of the inverse just written (each cell by one
    // thread; the waves of a workgroup share their compute unit's cache, the barrier orders the stores above)
    __syncthreads();
    const double sq = 0.70710678118654752440;
    for (int cell = tid; cell < h2 * h2; cell += BL_THREADS) {
      const int I = cell / h2, J = cell - I * h2;
      const bool ri = rotf[I] != 0, rj = rotf[J] != 0;
      if (!ri && !rj) continue;
      double* x00 = X + (size_t)I + (size_t)J * gs;
      double* x10 = X + (size_t)(I + h2) + (size_t)J * gs;
      double* x01 = X + (size_t)I + (size_t)(J + h2) * gs;
      double* x11 = X + (size_t)(I + h2) + (size_t)(J + h2) * gs;
      double a = *x00, b = *x01, c = *x10, d = *x11;
      if (ri) { const double t0 = sq * (a + c), t1 = sq * (a - c), u0 = sq * (b + d), u1 = sq * (b - d); a = t0; c = t1; b = u0; d = u1; }
      if (rj) { const double t0 = sq * (a + b), t1 = sq * (a - b), u0 = sq * (c + d), u1 = sq * (c - d); a = t0; b = t1; c = u0; d = u1; }
      *x00 = a; *x01 = b; *x10 = c; *x11 = d;
    }
    if ((gs & 1) && tid < h2 && rotf[tid]) {       // the unpaired last row and column against the rotated pairs
      const int I = tid, L = gs - 1;
      double* r0 = X + (size_t)I + (size_t)L * gs;
      double* r1 = X + (size_t)(I + h2) + (size_t)L * gs;
      double* c0 = X + (size_t)L + (size_t)I * gs;
      double* c1 = X + (size_t)L + (size_t)(I + h2) * gs;
      const double a = *r0, b = *r1, c = *c0, d = *c1;
      *r0 = sq * (a + b); *r1 = sq * (a - b); *c0 = sq * (c + d); *c1 = sq * (c - d);
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
  if (gs <= 16 * BL_NT) {
    // blocks of the unpivoted path (gs <= 112): the operands of ALL K steps are requested before the first product -- one
    // memory round trip per tile instead of one per 16 K steps (round 4: the level-0 products of C4 315 -> see DESIGN.md)
    double a[4 * BL_NT], b[4 * BL_NT];
#pragma unroll
    for (int u = 0; u < 4 * BL_NT; ++u) {
      const int k = 4 * u + lk;
      a[u] = (m < gs && k < gs) ? (TA ? A[(size_t)k + (size_t)m * gs] : A[(size_t)m + (size_t)k * gs]) : 0.0;
      b[u] = (n < gs && k < gs) ? (TB ? B[(size_t)n + (size_t)k * gs] : B[(size_t)k + (size_t)n * gs]) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4 * BL_NT; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
    return acc;
  }
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

// The three phases above in ONE launch per level (round 4), by pulling instead of pushing: workgroup e of ne + 1 owns the
// SURVIVING block j between the eliminated blocks i_lo = elim[e - 1] and i_up = elim[e] (ascending, 2 s apart: j = i_up - s =
// i_lo + s; the ends have one neighbour).  It forms u = inv b of both neighbours itself -- every u twice, by its two
// neighbours, bit for bit alike; the owner of i_up keeps it in w for the backward part -- and subtracts Klo^T u_up, then
// Kup u_lo from b_j: the order of the phases, hence the same sums.  Eliminated blocks' b are only read at their level.
// C4: 28 launches of 11 us -> 10 (the S solve is on the critical path of every time-staged back-solve).
__global__ __launch_bounds__(512) void k_bcr_fwd_fused(int gs, int G, BcrLevel lv, const double* __restrict__ inv,
                                                       const double* __restrict__ Klo, const double* __restrict__ Kup,
                                                       double* __restrict__ b, double* __restrict__ w) {
  __shared__ double part[8][512];
  __shared__ double ulo[512], t1[512];
  const int e = blockIdx.x, s = lv.s, r = threadIdx.x;
  const size_t g2 = (size_t)gs * gs;
  const int i_up = (e < lv.ne) ? lv.elim[e] : -1, i_lo = (e > 0) ? lv.elim[e - 1] : -1;
  const int j = (i_up >= 0) ? i_up - s : i_lo + s;
  const bool from_up = i_up >= 0 && lv.lo && j >= 0;                 // b_j -= Klo_{i_up}^T u_{i_up}
  const bool from_lo = i_lo >= 0 && i_lo + s == j && j < G;          // b_j -= Kup_{i_lo} u_{i_lo}
  if (i_up >= 0) {
    const double a = bcr_matvec(inv + (size_t)i_up * g2, b + (size_t)i_up * gs, gs, part);
    if (r < gs) w[(size_t)i_up * gs + r] = a;
  }
  if (from_lo) {
    const double a = bcr_matvec(inv + (size_t)i_lo * g2, b + (size_t)i_lo * gs, gs, part);
    if (r < gs) ulo[r] = a;
  }
  __syncthreads();                                                     // (w of this workgroup and ulo are read below)
  if (!from_up && !from_lo) return;
  double bj = (r < gs) ? b[(size_t)j * gs + r] : 0.0;
  if (from_up) {
    bcr_matvec_t(Klo + (size_t)i_up * g2, w + (size_t)i_up * gs, gs, t1);
    if (r < gs) bj -= t1[r];
  }
  if (from_lo) {
    const double a = bcr_matvec(Kup + (size_t)i_lo * g2, ulo, gs, part);
    if (r < gs) bj -= a;
  }
  if (r < gs) b[(size_t)j * gs + r] = bj;
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


}  // namespace

int ppi_btd_factor_schur(pp_handle h, const double* Q_host, long long corner_nnz) {
  hipStream_t st = h->stream;
  const int nc = h->nc;
  const size_t nn = schur_doubles(h);
  {
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
    if (pp::env_switch("PP_BCR_NO_LDS")) lds_bytes = 0;
    const bool bcr_mfma = pp::env_switch("PP_NO_BCR_MFMA") == nullptr;    // block products on the matrix cores, wave-level inverse (measurement switch)
    // unpivoted LDL^T + inverse of the blocks on the matrix cores, Bunch-Kaufman only for the blocks it rejects
    // (PP_NO_BCR_LDL: measurement switch; needs the matrix-core products for the rest of the level and gs <= 112)
    bool bcr_ldl = bcr_mfma && gs <= 16 * BL_NT && pp::env_switch("PP_NO_BCR_LDL") == nullptr;
    if (bcr_ldl && !h->bcr_ldl_attr) {
      if (hipFuncSetAttribute((const void*)k_bcr_ldl_inverse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BL_LDS_BYTES) != hipSuccess) {
        (void)hipGetLastError();
        bcr_ldl = false;
      } else {
        h->bcr_ldl_attr = true;
      }
    }
    int bk_threads = 256;        // (measured at C4, gs = 98, S phase per step: 64 threads 15.3 ms, 128 12.1, 256 10.8, 512 11.0)
    if (const char* e = pp::env_switch("PP_BCR_THREADS")) bk_threads = std::max(64, std::min(BK_THREADS, std::atoi(e)));
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
  return 0;
}

int ppi_btd_coupling_solve(pp_handle h, const double* rc_dev) {
  hipStream_t st = h->stream;
  const int nc = h->nc;
  {
    PhaseScope psb(h, 6, 1);
    {
      const int gs = h->gs, G = h->G, nlev = (int)h->bcr_ne.size();
      double* b = h->btd_vec + 2 * (size_t)nc + 16;
      double* w = b + nc + 16;
      hipLaunchKernelGGL(k_bcr_rhs, dim3((nc + 255) / 256), dim3(256), 0, st, nc, rc_dev, h->rs, b);
      for (int l = 0; l < nlev; ++l) {
        const BcrLevel lv{h->btd_elim + h->bcr_off[(size_t)l], h->bcr_ne[(size_t)l], h->bcr_s[(size_t)l], h->bcr_lo[(size_t)l]};
        static const bool three_phases = pp::env_switch("PP_BCR_FWD_PHASES") != nullptr;     // (measurement switch)
        if (!three_phases && gs <= 512)
          hipLaunchKernelGGL(k_bcr_fwd_fused, dim3(lv.ne + 1), dim3(512), 0, st, gs, G, lv, h->btd_inv, h->btd_klo, h->btd_kup, b, w);
        else
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
  return 0;
}
