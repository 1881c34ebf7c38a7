// Forward / backward substitution of the blocks and the coupling rows between them (mpi_explicit_schur_complement.py:381-396).
#include "common.hpp"
#include "kernels_transpose.hpp"

namespace {

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
// Chain fronts in the sweeps (plan.hpp, PlanOptions::chain_sweeps).  The columns of a front form ONE level: what they
// receive from panels outside the front comes through the ordinary row tasks of that level (forward: k_fwd_level over the
// outside entries only), the dependencies between the front's own panels -- one sweep level each without these kernels:
// 10-16 launches at their floor per front -- are walked inside one workgroup per (front, chunk of 64 instances), lane =
// instance, the front's part of the vector in LDS.
// Forward: y of panel i is final once the earlier panels are applied; then every later column c of the front gets
// y_c -= sum_k L_i[row of c][k] y_(i, k) (its waves take the columns NW apart).
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_chain_fwd(GroupDev g, int front0, int ny) {
  __shared__ double yl[64][64];                     // [front column][lane]   (PlanOptions::chain_wmax <= 64)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fi = front0 + (int)(blockIdx.x / (unsigned)ny);
  const unsigned b = (unsigned)((blockIdx.x % (unsigned)ny) * 64 + lane);
  const size_t bpad = (size_t)g.bpad;
  const int* H = g.chain_hdr + 8 * (size_t)fi;
  const int W = H[1], npan = H[2];
  const int* PR = g.chain_pan + 8 * (size_t)H[3];
  const int* col = g.chain_col + H[4];
  for (int c = wave; c < W; c += NW) yl[c][lane] = g.Y[(size_t)col[c] * bpad + b];
  __syncthreads();
  for (int i = 0; i + 1 < npan; ++i) {
    const int* R = PR + 8 * i;
    const int w = R[1], c0 = R[6], c1 = c0 + w;
    const double* Lp = g.L + (size_t)R[2] * bpad + b;
    double yk[PP_WMAX];
#pragma unroll
    for (int k = 0; k < PP_WMAX; ++k) yk[k] = (k < w) ? yl[c0 + min(k, w - 1)][lane] : 0.0;
    // (the operands of four columns requested together: one round trip instead of four)
    for (int cb = c1 + wave; cb < W; cb += 4 * NW) {
      double l[4][PP_WMAX];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int t = min(cb + j * NW, W - 1) - c0;     // the row of the column in panel i
#pragma unroll
        for (int k = 0; k < PP_WMAX; ++k) l[j][k] = Lp[(size_t)(t * w + min(k, w - 1)) * bpad];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = cb + j * NW;
        if (c < W) {
          double acc = yl[c][lane];
#pragma unroll
          for (int k = 0; k < PP_WMAX; ++k)
            if (k < w) acc -= l[j][k] * yk[k];
          yl[c][lane] = acc;
        }
      }
    }
    __syncthreads();
  }
  for (int c = wave; c < W; c += NW) g.Y[(size_t)col[c] * bpad + b] = yl[c][lane];
}

// Backward: the ordinary column tasks of the level have left  x_c = (inv(P_p) y_p)_q - sum over the rows BELOW THE FRONT'S
// PIVOTS of L_p[t][q] x(row t)  in X for every column of the front (their records name only those rows: the bulk of the
// panel, streamed by one workgroup per column across the chip); what is left is the triangular part inside the front.  The
// panels are taken in reverse: x_(p, q) -= sum over the rows t of panel p that are later columns of the front of
// L_p[t][q] x(that column) -- the rows spread over the waves, partial sums through LDS in wave order (deterministic), wave q
// finishes column q.  LDS: xl[W][64], red[NW][PP_WMAX][64].
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_chain_bwd(GroupDev g, int front0, int ny, const int* __restrict__ colX) {
  __shared__ double xl[64][64];                     // [front column][lane]   (PlanOptions::chain_wmax <= 64)
  __shared__ double red[NW][PP_WMAX][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fi = front0 + (int)(blockIdx.x / (unsigned)ny);
  const int b = (int)((blockIdx.x % (unsigned)ny) * 64 + lane);
  const size_t bpad = (size_t)g.bpad;
  const int* H = g.chain_hdr + 8 * (size_t)fi;
  const int W = H[1], npan = H[2];
  const int* PR = g.chain_pan + 8 * (size_t)H[3];
  const int* cx = colX + H[4];
  for (int c = wave; c < W; c += NW) xl[c][lane] = g.X[(size_t)cx[c] * bpad + b];
  __syncthreads();
  for (int i = npan - 2; i >= 0; --i) {             // (the last panel has no later column of the front below it)
    const int* R = PR + 8 * i;
    const int w = R[1], c0 = R[6], nint = W - c0;   // rows [w, nint) of the panel are the later columns of the front
    const double* Lp = g.L + (size_t)R[2] * bpad + b;
    double s[PP_WMAX];
#pragma unroll
    for (int q = 0; q < PP_WMAX; ++q) s[q] = 0.0;
    // (the rows of a wave four at a time: sixteen operand requests in flight)
    for (int tb = w + wave; tb < nint; tb += 4 * NW) {
      double l[4][PP_WMAX], xv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int t = min(tb + j * NW, nint - 1);
#pragma unroll
        for (int q = 0; q < PP_WMAX; ++q) l[j][q] = Lp[(size_t)(t * w + min(q, w - 1)) * bpad];
        xv[j] = (tb + j * NW < nint) ? xl[c0 + t][lane] : 0.0;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < PP_WMAX; ++q) s[q] += l[j][q] * xv[j];
    }
#pragma unroll
    for (int q = 0; q < PP_WMAX; ++q) red[wave][q][lane] = s[q];
    __syncthreads();
    if (wave < w) {
      const int q = wave;
      double tot = 0.0;
      for (int j = 0; j < NW; ++j) tot += red[j][q][lane];
      const double x = (b < g.batch) ? xl[c0 + q][lane] - tot : 0.0;
      xl[c0 + q][lane] = x;
      g.X[(size_t)cx[c0 + q] * bpad + b] = x;
    }
    __syncthreads();
  }
}

// per-instance copy of the coupling solution for the back substitution of a mapped group
__global__ __launch_bounds__(256) void k_gather_xc(GroupDev g, const double* __restrict__ xc) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)g.nc * g.bpad) return;
  const int b = (int)(i % g.bpad);
  g.XCL[i] = (b < g.batch) ? xc[g.cmapT[i]] : 0.0;
}


}  // namespace

extern "C" {

int pp_solve_forward(pp_handle h) { return pp_solve_forward_ex(h, 0); }

// rhs_before_factor != 0: the caller states that the bound right-hand side was complete (in the handle's stream order)
// BEFORE the block factorisation of this step was enqueued (an interior-point iteration: pp_bind_native_vectors, then
// pp_numeric_local ...).  With pattern groups on streams of their own and a block-tridiagonal S the sweep of a group then
// starts behind the factorisation of ITS group on ITS stream -- not behind the Schur update and the cyclic reduction the
// handle's stream holds by now (C4: 0.8 + 1.4 ms per step, against 1.3 ms of forward sweeps) -- and the groups of the
// handle's own stream move to the auxiliary stream with the fewest levels, behind the event of the factorisation's join.
// No additional stream: a fourth one made the three of C4 share a hardware queue (api.hip, pp_end_symbolic).
int pp_solve_forward_ex(pp_handle h, int rhs_before_factor) {
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
  const bool early = rhs_before_factor && h->btd && h->blocks_done_valid && h->fwd_early && !h->profile && h->nsplit_req <= 1 &&
                     h->group_streams && h->groups.size() > 1 && h->aux_made;
  if (early) {
    // (no fork event: every auxiliary stream already holds the factorisation of its groups)
    gst.n = (int)std::min<size_t>(h->groups.size(), (size_t)PP_MAX_SPLIT);
    gst.st[0] = h->stream;
    for (int i = 1; i < gst.n; ++i) gst.st[i] = h->aux[i];
  } else if (fork_group_streams(h, gst)) {
    return fail(h, 3, "stream fork failed");
  }
  auto group_body_on = [&](size_t gi, hipStream_t st) -> int {
    Group* g = h->groups[gi];
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
        if (P.chain_sweeps_on && P.chain_lvl_ptr[l + 1] > P.chain_lvl_ptr[l]) {
          if (sp.n != 1) return fail(h, 3, "instance splits are not supported together with chain fronts");
          hipLaunchKernelGGL(k_chain_fwd<8>, dim3((unsigned)(P.chain_lvl_ptr[l + 1] - P.chain_lvl_ptr[l]) * (unsigned)d.nchunk),
                             dim3(64 * 8), 0, fan[0], dn, P.chain_lvl_ptr[l], d.nchunk);
        }
      }
      if (join_streams(h, sp, fan)) return fail(h, 3, "stream join failed");
    }
    return 0;
  };
  if (!early) {
    auto group_body = [&](size_t gi) -> int { return group_body_on(gi, gst.st[gi % (size_t)gst.n]); };
    if (int rc = run_groups(h, gst, group_body)) return rc;
  } else {
    const size_t ng = h->groups.size();
    const int n = gst.n;
    int host = 1, fewest = 1 << 30;          // the auxiliary stream that also takes the groups of the handle's stream
    for (int k = 1; k < n; ++k) {
      int lv = 0;
      for (size_t gi = (size_t)k; gi < ng; gi += (size_t)n) lv += h->groups[gi]->plan.n_levels;
      if (lv < fewest) { fewest = lv; host = k; }
    }
    int rcs[PP_MAX_SPLIT] = {0};
    auto stream_body = [&](int k) {
      if (k == 0) return;
      for (size_t gi = (size_t)k; gi < ng; gi += (size_t)n)
        if (int rc = group_body_on(gi, gst.st[k])) { rcs[k] = rc; return; }
      if (k != host) return;
      if (hipStreamWaitEvent(gst.st[k], h->ev_blocks_done, 0) != hipSuccess) { rcs[k] = 3; return; }
      for (size_t gi = 0; gi < ng; gi += (size_t)n)
        if (int rc = group_body_on(gi, gst.st[k])) { rcs[k] = rc; return; }
    };
    if (h->enqueue_threads) h->pool.run(n, h->device, stream_body);
    else for (int k = 0; k < n; ++k) stream_body(k);
    for (int k = 0; k < n; ++k)
      if (rcs[k]) return rcs[k] == 3 && h->err.empty() ? fail(h, 3, "early forward sweep: enqueue failed") : rcs[k];
  }
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

int pp_solve_coupling_dev(pp_handle h, const double* rc_dev);

int pp_solve_coupling(pp_handle h, const double* rc_host) {
  if (!h || !h->schur_done) return fail(h, 3, "pp_solve_coupling before pp_factor_schur");
  PP_HIP(hipSetDevice(h->device));
  const int nc = h->nc;
  if (nc == 0) return 0;
  if (rc_host) PP_HIP(hipMemcpyAsync(h->rcd, rc_host, (size_t)nc * sizeof(double), hipMemcpyHostToDevice, h->stream));
  return pp_solve_coupling_dev(h, rc_host ? h->rcd : nullptr);
}

int pp_solve_coupling_dev(pp_handle h, const double* rc_dev) {
  if (!h || !h->schur_done) return fail(h, 3, "pp_solve_coupling_dev before pp_factor_schur");
  PP_HIP(hipSetDevice(h->device));
  if (h->nc == 0) return 0;
  if (!h->refining) h->last_rc = rc_dev;      // (the a-posteriori check measures the coupling rows against it: refine.hip)
  return h->btd ? ppi_btd_coupling_solve(h, rc_dev) : ppi_dense_coupling_solve(h, rc_dev);      // bcr.hip / dense.hip
}

double* pp_coupling_solution_buffer(pp_handle h) { return (h && h->symbolic_done) ? h->xc : nullptr; }

int pp_copy_coupling_solution(pp_handle h, double* dev_ptr) {
  if (!h || !h->schur_done) return fail(h, 3, "pp_copy_coupling_solution before pp_factor_schur");
  PP_HIP(hipSetDevice(h->device));
  if (h->nc > 0) PP_HIP(hipMemcpyAsync(dev_ptr, h->xc, (size_t)h->nc * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
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
        const int c0 = P.clevel_ptr[l];
        const int ncol = P.clevel_ptr[l + 1] - c0;
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
        if (P.chain_sweeps_on && P.chain_lvl_ptr[l + 1] > P.chain_lvl_ptr[l]) {
          // (the columns of the level's fronts have their part from below the fronts: now the triangle inside each front)
          if (sp.n != 1) return fail(h, 3, "instance splits are not supported together with chain fronts");
          hipLaunchKernelGGL(k_chain_bwd<8>, dim3((unsigned)(P.chain_lvl_ptr[l + 1] - P.chain_lvl_ptr[l]) * (unsigned)d.nchunk),
                             dim3(64 * 8), 0, fan[0], dn, P.chain_lvl_ptr[l], d.nchunk, native ? g->chain_colN : d.chain_col);
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

int pp_get_coupling_solution(pp_handle h, double* xc_host) {
  if (!h || !h->schur_done) return fail(h, 3, "pp_get_coupling_solution before pp_factor_schur");
  PP_HIP(hipSetDevice(h->device));
  if (h->nc > 0) PP_HIP(hipMemcpyAsync(xc_host, h->xc, (size_t)h->nc * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  PP_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

}  // extern "C"
