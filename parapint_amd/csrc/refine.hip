// A-posteriori check of a back-solve and iterative refinement (round 6).
//
// The reference's sub-solvers pivot every block on its own values (MA27: parapint/linalg/ma27_interface.py:36-47, 110-140;
// SuperLU: scipy_interface.py:26-31) -- a back-solve of theirs is backward stable whatever the values.  The batched
// factorisation here fixes ONE pivot sequence per pattern group (plan.hpp), so the accuracy of a solve is checked where
// it can be seen: r_i = b_i - K_i x_i - A_i^T x_c for every local block, from the values the factorisation read (the
// transposed input, or the producer's source arrays through the value map), per instance as
//     rho_b = max_rows |r| / max_rows (sum_j |K_ij| |x_j| + |A^T x_c| + |b|)      (a row-wise backward error),
// the worst instance published to a pinned mailbox.  The coupling rows b_c - sum_i A_i x_i - Q x_c hold in exact arithmetic
// whatever the block factors are (S, r_s and the backward sweep use the SAME factors), but an unstable pivot sequence
// amplifies their rounding errors: this unit forms sum_i A_i x_i and sum_i |A_i| |x_i| over the local blocks and hands them
// (with x_c and b_c) to the host class through the same mailbox, which adds Q x_c -- a few n_c-vectors -- after the
// all-reduce that also agrees the verdict between the ranks.  The host class (hip_schur_complement.py, solution_check.py)
// reads rho after every back-solve, runs correction solves K d = r through the same sweeps while rho is above its
// refinement threshold, and treats a solve that stays above 1e-8 as a breakdown of the pivot sequence (new sequence
// from that instance, factorise, solve again) -- never as a result.
//
// Layout: rows in the plan's elimination order, one wave = ROWS consecutive rows x 64 (or 128) instances; the records
// {value row, x row} are wave-uniform (scalar loads), every operand access is a coalesced 512-byte (1 KB) request.
// HBM-bound: every read value of K once per triangle it belongs to (C3: 36 k records per block = 301 MB of operand
// requests per 1024 blocks, of which the values are 167 MB unique), x and b once plus the gathers the L2 serves.
#include "common.hpp"
#include "kernels_transpose.hpp"

namespace {

constexpr int RES_NW = 8;           // waves per workgroup: their per-instance maxima meet in LDS, one atomic per lane and workgroup

__device__ __forceinline__ unsigned long long dbits(double v) { return (unsigned long long)__double_as_longlong(v); }

// NV instances per lane (as in factor.hip).  vrow < 0: the constant 1 (value-map form: source row == const_row is
// rewritten to -1 at build time).  xcol >= 0: row of X; xcol < 0: coupling value -1 - xcol (xc / per-instance XCL).
// Arguments of the residual launch that do not fit a plain parameter list comfortably
struct ResArgs {
  const int *rptr, *vrow, *xcol, *brow, *rrow;       // rows of [K | A^T]: entry ranges, value rows, x rows, b rows, rows of Rout
  const double* coef;                                // value-map form: coefficient per record (else null = 1)
  const int *bptr, *bvrow, *bxcol;                   // the border by coupling row
  const double* bcoef;
  const double *V, *B, *X, *xc;
  double* Rout;
  unsigned long long *rmax, *smax;
  double *bpart, *ax;                                // border partial sums [chunk][2][nc] (uniform groups) / coupling sums (mapped)
  int nres_wg, ny, rows_per_wg, nc_glob, nrows;      // nrows: rows the check evaluates (positions of the execution order)
};

// NV instances per lane (as in factor.hip).  vrow < 0: the constant 1.  xcol >= 0: row of X; xcol < 0: coupling value
// -1 - xcol (xc / per-instance XCL).  The workgroups behind the residual rows take the coupling rows: per (local coupling
// row c, chunk) one wave forms sum_e A[c, j_e] x_(j_e) and the same with absolute values over its 64 instances; uniform
// groups reduce over the lanes into bpart[chunk][2][nc] (summed by k_residual_reduce), mapped groups add every instance's
// term to its own global coupling row.
template <int NV, bool STORE>
__global__ __launch_bounds__(64 * RES_NW) void k_residual(GroupDev g, ResArgs a, const int* __restrict__ rptr, const int* __restrict__ vrow,
                                                 const int* __restrict__ xcol, const int* __restrict__ brow,
                                                 const double* __restrict__ coef, const double* __restrict__ V,
                                                 const double* __restrict__ B, const double* __restrict__ X,
                                                 const double* __restrict__ xc) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t bpad = (size_t)g.bpad;
  if ((int)blockIdx.x >= a.nres_wg) {
    const int t = ((int)blockIdx.x - a.nres_wg) * RES_NW + wave;
    if (t >= g.nc * g.nchunk) return;
    const int chunk = t % g.nchunk, c = t / g.nchunk;
    const int b = chunk * 64 + lane;
    double s = 0.0, ab = 0.0;
    for (int e = a.bptr[c]; e < a.bptr[c + 1]; ++e) {
      const int vr = a.bvrow[e];
      const double tm = (a.bcoef ? a.bcoef[e] : 1.0) * (vr >= 0 ? a.V[(size_t)vr * bpad + b] : 1.0) * a.X[(size_t)a.bxcol[e] * bpad + b];
      s += tm; ab += fabs(tm);
    }
    if (b >= g.batch) { s = 0.0; ab = 0.0; }
    if (g.cmapT) {
      if (b < g.batch && (s != 0.0 || ab != 0.0)) {
        const int r = g.cmapT[(size_t)c * bpad + b];
        atomicAdd(a.ax + r, s); atomicAdd(a.ax + a.nc_glob + r, ab);
      }
      return;
    }
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); ab += __shfl_xor(ab, off); }
    if (lane == 0) { a.bpart[((size_t)chunk * 2) * g.nc + c] = s; a.bpart[((size_t)chunk * 2 + 1) * g.nc + c] = ab; }
    return;
  }
  __shared__ double red[RES_NW][2 * NV][64];
  const unsigned b = (unsigned)(((blockIdx.x % (unsigned)a.ny) * 64 + lane) * NV);
  const int c0 = ((int)(blockIdx.x / (unsigned)a.ny) * RES_NW + wave) * a.rows_per_wg;
  const int c1 = min(c0 + a.rows_per_wg, a.nrows);
  // (the records as restrict-qualified kernel parameters, and no store inside the row loop unless STORE: through the
  // argument struct, or with stores in between, the compiler fetches every record with a vector load + v_readfirstlane
  // instead of a scalar load -- measured 125 against 95 us at C3)
  const double* __restrict__ Vb = V + b;
  const double* __restrict__ Xb = X + b;
  const double* __restrict__ Bb = B + b;
  double* __restrict__ Rout = STORE ? a.Rout + b : nullptr;
  double rm[NV], sm[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) { rm[v] = 0.0; sm[v] = 0.0; }
  for (int c = c0; c < c1; ++c) {
    const int e0 = rptr[c], e1 = rptr[c + 1];
    double bv[NV];
    ldv<NV>(Bb + (size_t)(brow ? brow[c] : c) * bpad, bv);
    double acc[NV], aab[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) { acc[v] = 0.0; aab[v] = 0.0; }
    for (int eb = e0; eb < e1; eb += 4) {
      double kv[4][NV], xv[4][NV], cf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = min(eb + i, e1 - 1);
        const int vr = vrow[e], xr = xcol[e];
        cf[i] = (eb + i < e1) ? (coef ? coef[e] : 1.0) : 0.0;
        if (vr >= 0) ldv<NV>(Vb + (size_t)vr * bpad, kv[i]);
        else {
#pragma unroll
          for (int v = 0; v < NV; ++v) kv[i][v] = 1.0;
        }
        if (xr >= 0) ldv<NV>(Xb + (size_t)xr * bpad, xv[i]);
        else {
#pragma unroll
          for (int v = 0; v < NV; ++v) xv[i][v] = xc[(size_t)(-1 - xr) * g.xs_row + (size_t)(b + v) * g.xs_lane];
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const double t = cf[i] * kv[i][v] * xv[i][v];
          acc[v] += t;
          aab[v] += fabs(t);
        }
      }
    }
    double r[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const bool live = (int)(b + v) < g.batch;
      r[v] = live ? bv[v] - acc[v] : 0.0;
      // (a NaN anywhere in x makes |r| NaN: fmax would drop it, so it is turned into +inf here -- never a silent pass)
      const double ar = (r[v] == r[v]) ? fabs(r[v]) : INFINITY;
      rm[v] = fmax(rm[v], ar);
      sm[v] = fmax(sm[v], live ? aab[v] + fabs(bv[v]) : 0.0);
    }
    if (STORE) stv<NV>(Rout + (size_t)(a.rrow ? a.rrow[c] : c) * bpad, r);
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) { red[wave][v][lane] = rm[v]; red[wave][NV + v][lane] = sm[v]; }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    for (int k = 1; k < RES_NW; ++k) { rm[v] = fmax(rm[v], red[k][v][lane]); sm[v] = fmax(sm[v], red[k][NV + v][lane]); }
    if ((int)(b + v) < g.batch) {
      // (non-negative doubles order like their bit patterns; +inf is the largest)
      atomicMax(a.rmax + b + v, dbits(rm[v]));
      atomicMax(a.smax + b + v, dbits(sm[v]));
    }
  }
}

// One workgroup per group, one after the other on the handle's stream: rho_b = rmax_b / smax_b (0 / 0 = 0), the worst
// instance of the group against the best so far in best = {rho, group, slot, largest row scale}; the per-instance maxima
// are reset; the group's border sums are added to the handle's coupling sums ax | aabs (uniform groups: from the per-chunk
// partials; the first group stores).  The LAST group's launch finishes the check:
//   coupling rows on the device (Qmode >= 0: dense S, one rank -- all sums are complete here):
//     r_c = b_c - sum A x - Q x_c  (Qmode 1: Q dense column-major, lower triangle read; 0: Q = 0), measured against
//     max(row scales of the coupling rows, largest row scale of the blocks) -> mail {rho, group, slot, seq, scale, rho_c};
//     rc_out (store): r_c for the correction solve;
//   else (Qmode < 0: several ranks or a block-tridiagonal S -- the caller finishes the coupling rows after its all-reduce):
//     the header only; x_c, the sums and b_c follow by copies behind this launch, and k_residual_flag signals.
__global__ __launch_bounds__(256) void k_residual_reduce(unsigned long long* __restrict__ rmax, unsigned long long* __restrict__ smax,
                                                         int batch, int gid, int first, int last, double* __restrict__ best,
                                                         double* mail, long long seq, int nc, int nc_loc, int nchunk,
                                                         const double* __restrict__ bpart, double* __restrict__ ax,
                                                         const double* __restrict__ xc, const double* __restrict__ bc, int Qmode,
                                                         const double* __restrict__ Q, double* __restrict__ rc_out) {
  __shared__ double srho[256], sscale[256];
  __shared__ int sslot[256];
  double rho = 0.0, scale = 0.0;
  int slot = -1;
  for (int b = threadIdx.x; b < batch; b += 256) {
    const double r = __longlong_as_double((long long)rmax[b]), s = __longlong_as_double((long long)smax[b]);
    rmax[b] = 0ull; smax[b] = 0ull;
    scale = fmax(scale, s);
    const double q = (r == 0.0) ? 0.0 : ((s > 0.0 && r == r) ? r / s : INFINITY);
    if (q > rho || slot < 0) { rho = q; slot = b; }
  }
  if (bpart) {          // uniform group: its border sums (local row = global row)
    for (int c = threadIdx.x; c < nc_loc; c += 256) {
      double s = 0.0, ab = 0.0;
#pragma unroll 8
      for (int q = 0; q < nchunk; ++q) { s += bpart[((size_t)q * 2) * nc_loc + c]; ab += bpart[((size_t)q * 2 + 1) * nc_loc + c]; }
      ax[c] = first ? s : ax[c] + s;
      ax[nc + c] = first ? ab : ax[nc + c] + ab;
    }
  }
  srho[threadIdx.x] = rho; sslot[threadIdx.x] = slot; sscale[threadIdx.x] = scale;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      sscale[threadIdx.x] = fmax(sscale[threadIdx.x], sscale[threadIdx.x + off]);
      const double o = srho[threadIdx.x + off];
      const int os = sslot[threadIdx.x + off];
      // (ties go to the lower slot: deterministic)
      if (os >= 0 && (sslot[threadIdx.x] < 0 || o > srho[threadIdx.x] || (o == srho[threadIdx.x] && os < sslot[threadIdx.x]))) {
        srho[threadIdx.x] = o; sslot[threadIdx.x] = os;
      }
    }
    __syncthreads();
  }
  __shared__ double bscale;
  if (threadIdx.x == 0) {
    const double brho = first ? -1.0 : best[0];
    if (srho[0] > brho) { best[0] = srho[0]; best[1] = (double)gid; best[2] = (double)sslot[0]; }
    best[3] = first ? sscale[0] : fmax(best[3], sscale[0]);       // largest row scale |K||x| + |b| of the local blocks
    bscale = best[3];
  }
  __syncthreads();
  if (!last || Qmode == -2) return;
  double rho_c = 0.0;
  if (Qmode >= 0 && nc > 0) {
    double rmx = 0.0, smx = 0.0;
    for (int c = threadIdx.x; c < nc; c += 256) {
      const double bcv = bc ? bc[c] : 0.0;
      double r = bcv - ax[c], s = fabs(bcv) + ax[nc + c];
      if (Qmode == 1) {
        for (int k = 0; k < nc; ++k) {
          const double q = (c >= k) ? Q[(size_t)c + (size_t)k * nc] : Q[(size_t)k + (size_t)c * nc];
          r -= q * xc[k];
          s += fabs(q) * fabs(xc[k]);
        }
      }
      if (rc_out) rc_out[c] = r;
      rmx = fmax(rmx, (r == r) ? fabs(r) : INFINITY);
      smx = fmax(smx, s);
    }
    __syncthreads();
    srho[threadIdx.x] = rmx; sscale[threadIdx.x] = smx;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) {
        srho[threadIdx.x] = fmax(srho[threadIdx.x], srho[threadIdx.x + off]);
        sscale[threadIdx.x] = fmax(sscale[threadIdx.x], sscale[threadIdx.x + off]);
      }
      __syncthreads();
    }
    const double den = fmax(sscale[0], bscale);
    rho_c = (srho[0] == 0.0) ? 0.0 : ((den > 0.0 && srho[0] < INFINITY) ? srho[0] / den : INFINITY);
  }
  if (threadIdx.x == 0) {
    mail[0] = best[0]; mail[1] = best[1]; mail[2] = best[2]; mail[4] = best[3]; mail[5] = rho_c; mail[6] = best[0];
    if (Qmode >= 0) {
      __threadfence_system();
      __hip_atomic_store(reinterpret_cast<long long*>(mail) + 3, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Several ranks, the library's communicator: this rank's block result into its slot pair behind the coupling sums (the
// other slots zero), so that ONE sum all-reduce carries the sums of the coupling rows and every rank's block result ...
__global__ __launch_bounds__(64) void k_residual_slots(const double* __restrict__ best, double* __restrict__ ax, int nc, int nranks, int rank) {
  for (int r = threadIdx.x; r < nranks; r += 64) {
    ax[2 * (size_t)nc + 2 * r] = (r == rank) ? fmin(best[0], 1e300) : 0.0;
    ax[2 * (size_t)nc + 2 * r + 1] = (r == rank) ? fmin(best[3], 1e300) : 0.0;
  }
}

// ... and every rank finishes identically from the all-reduced buffer: the worst block of any rank, the coupling rows
// r_c = b_c - sum A x - Q x_c against max(their row scales, the largest block row scale of any rank).
__global__ __launch_bounds__(256) void k_residual_finish(const double* __restrict__ best, double* mail, long long seq, int nc, int nranks,
                                                         const double* __restrict__ ax, const double* __restrict__ xc,
                                                         const double* __restrict__ bc, int Qmode, const double* __restrict__ Q,
                                                         double* rc_out, const double* s_pre) {
  __shared__ double sr[256], ss[256];
  double rall = nranks > 0 ? 0.0 : best[0], sall = nranks > 0 ? 0.0 : best[3];      // (one rank: its own block result)
  for (int r = threadIdx.x; r < nranks; r += 256) { rall = fmax(rall, ax[2 * (size_t)nc + 2 * r]); sall = fmax(sall, ax[2 * (size_t)nc + 2 * r + 1]); }
  double rmx = 0.0, smx = 0.0;
  for (int c = threadIdx.x; c < nc; c += 256) {
    const double bcv = bc ? bc[c] : 0.0;
    // (Qmode 2, sparse Q of a block-tridiagonal S: k_corner_rc_init / _apply have formed r and its scale in rc_out / s_pre)
    double r = (Qmode == 2) ? rc_out[c] : bcv - ax[c], s = (Qmode == 2) ? s_pre[c] : fabs(bcv) + ax[nc + c];
    if (Qmode == 1) {
      for (int k = 0; k < nc; ++k) {
        const double q = (c >= k) ? Q[(size_t)c + (size_t)k * nc] : Q[(size_t)k + (size_t)c * nc];
        r -= q * xc[k];
        s += fabs(q) * fabs(xc[k]);
      }
    }
    if (rc_out && Qmode != 2) rc_out[c] = r;
    rmx = fmax(rmx, (r == r) ? fabs(r) : INFINITY);
    smx = fmax(smx, s);
  }
  sr[threadIdx.x] = rall; ss[threadIdx.x] = sall;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) { sr[threadIdx.x] = fmax(sr[threadIdx.x], sr[threadIdx.x + off]); ss[threadIdx.x] = fmax(ss[threadIdx.x], ss[threadIdx.x + off]); }
    __syncthreads();
  }
  const double rho_all = sr[0], scale_all = ss[0];
  __syncthreads();
  sr[threadIdx.x] = rmx; ss[threadIdx.x] = smx;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) { sr[threadIdx.x] = fmax(sr[threadIdx.x], sr[threadIdx.x + off]); ss[threadIdx.x] = fmax(ss[threadIdx.x], ss[threadIdx.x + off]); }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double den = fmax(ss[0], scale_all);
    const double rho_c = (sr[0] == 0.0) ? 0.0 : ((den > 0.0 && sr[0] < INFINITY) ? sr[0] / den : INFINITY);
    mail[0] = best[0]; mail[1] = best[1]; mail[2] = best[2]; mail[4] = scale_all; mail[5] = rho_c; mail[6] = rho_all;
    __threadfence_system();
    __hip_atomic_store(reinterpret_cast<long long*>(mail) + 3, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Block-tridiagonal S: Q as the (position in the Schur layout, value) pairs pp_factor_schur_corner got -- D[G][gs][gs] (both
// triangles of a diagonal block are listed) | E[G-1][gs][gs] with E_t = S(block t + 1, block t) listed once.
__global__ __launch_bounds__(256) void k_corner_rc_init(int nc, const double* __restrict__ bc, const double* __restrict__ ax,
                                                        double* __restrict__ r, double* __restrict__ s) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= nc) return;
  const double bcv = bc ? bc[c] : 0.0;
  r[c] = bcv - ax[c];
  s[c] = fabs(bcv) + ax[nc + c];
}
__global__ __launch_bounds__(256) void k_corner_rc_apply(long long nnz, const long long* __restrict__ pos, const double* __restrict__ val,
                                                         int gs, int G, const double* __restrict__ xc, double* __restrict__ r,
                                                         double* __restrict__ s) {
  const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
  if (k >= nnz) return;
  const long long g2 = (long long)gs * gs, p = pos[k];
  const double v = val[k];
  if (v == 0.0) return;
  if (p < (long long)G * g2) {
    const int t = (int)(p / g2), c = (int)((p % g2) / gs), rr = (int)(p % gs);
    const int row = t * gs + rr, col = t * gs + c;
    atomicAdd(r + row, -v * xc[col]); atomicAdd(s + row, fabs(v) * fabs(xc[col]));
  } else {
    const long long q = p - (long long)G * g2;
    const int t = (int)(q / g2), c = (int)((q % g2) / gs), rr = (int)(q % gs);
    const int row = (t + 1) * gs + rr, col = t * gs + c;
    atomicAdd(r + row, -v * xc[col]); atomicAdd(s + row, fabs(v) * fabs(xc[col]));
    atomicAdd(r + col, -v * xc[row]); atomicAdd(s + col, fabs(v) * fabs(xc[row]));
  }
}

__global__ void k_residual_flag(double* mail, long long seq) {
  __threadfence_system();
  __hip_atomic_store(reinterpret_cast<long long*>(mail) + 3, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// dst[rd(c)][b] += src[rs(c)][b]  (c < n; null maps = identity) -- the correction of a refinement step
__global__ __launch_bounds__(256) void k_add_rows(double* __restrict__ dst, const int* __restrict__ rd,
                                                  const double* __restrict__ src, const int* __restrict__ rs, int n, int bpad) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)n * bpad) return;
  const int c = (int)(i / (size_t)bpad), b = (int)(i % (size_t)bpad);
  dst[(size_t)(rd ? rd[c] : c) * bpad + b] += src[(size_t)(rs ? rs[c] : c) * bpad + b];
}

__global__ __launch_bounds__(256) void k_add_vec(double* __restrict__ dst, const double* __restrict__ src, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

__global__ __launch_bounds__(256) void k_gather_xc_refine(GroupDev g, const double* __restrict__ xc) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)g.nc * g.bpad) return;
  const int b = (int)(i % g.bpad);
  g.XCL[i] = (b < g.batch) ? xc[g.cmapT[i]] : 0.0;
}

}  // namespace

// Records of the residual rows of one group (api.hip: pp_end_symbolic).  Row c (elimination order) of
// [K_i | A_i^T]: every canonical lower entry (i, j) of K gives x_j to row i and, off the diagonal, x_i to row j; every
// border entry (coupling row r, column j) gives xc_r to row j.  A canonical entry is the sum of its raw entries.
int ppi_build_residual_records(pp_handle h, Group* g, const std::vector<int>& rawmap) {
  const pp::Plan& P = g->plan;
  const int n = P.n;
  const int32_t *rowK = g->pat_rowK.data(), *colK = g->pat_colK.data(), *rowB = g->pat_rowB.data(), *colB = g->pat_colB.data();
  const int nnzK = (int)g->pat_rowK.size(), nnzB = (int)g->pat_rowB.size();
  std::vector<std::vector<std::array<int, 2>>> rows((size_t)n);      // {canonical entry, x column (new) or -1 - coupling row}
  for (int e = 0; e < nnzK; ++e) {
    const int i = P.iperm[(size_t)rowK[e]], j = P.iperm[(size_t)colK[e]];
    rows[(size_t)i].push_back({e, j});
    if (i != j) rows[(size_t)j].push_back({e, i});
  }
  for (int e = 0; e < nnzB; ++e) rows[(size_t)P.iperm[(size_t)colB[e]]].push_back({nnzK + e, -1 - rowB[e]});
  // Order of execution: reverse Cuthill-McKee on the graph of K.  Every off-diagonal value serves two rows (i and j), every
  // x_j the rows of j's neighbours: with the two rows in one task or in neighbouring ones (which run on the same XCD at
  // about the same time) the second request meets the line in the L2.  MEASURED at C3 (rocprofv3 --pmc FETCH_SIZE):
  // elimination order 500 MB past the L2 per launch against 281 MB of distinct operands.
  std::vector<int> order;
  {
    order.reserve((size_t)n);
    std::vector<int> deg((size_t)n, 0);
    for (int c = 0; c < n; ++c) deg[(size_t)c] = (int)rows[(size_t)c].size();
    std::vector<char> seen((size_t)n, 0);
    std::vector<int> by_deg((size_t)n);
    for (int c = 0; c < n; ++c) by_deg[(size_t)c] = c;
    std::stable_sort(by_deg.begin(), by_deg.end(), [&](int a2, int b2) { return deg[(size_t)a2] < deg[(size_t)b2]; });
    std::vector<int> nb;
    for (int start : by_deg) {
      if (seen[(size_t)start]) continue;
      size_t head = order.size();
      order.push_back(start); seen[(size_t)start] = 1;
      while (head < order.size()) {
        const int c = order[head++];
        nb.clear();
        for (auto& rc : rows[(size_t)c]) if (rc[1] >= 0 && !seen[(size_t)rc[1]]) { seen[(size_t)rc[1]] = 1; nb.push_back(rc[1]); }
        std::stable_sort(nb.begin(), nb.end(), [&](int a2, int b2) { return deg[(size_t)a2] < deg[(size_t)b2]; });
        order.insert(order.end(), nb.begin(), nb.end());
      }
    }
    std::reverse(order.begin(), order.end());
  }
  // Rows that hold by construction are not evaluated: a 1 x 1 pivot column c without incoming entries (a leaf of the
  // elimination tree: d_c = K_cc, l_ic = K_ic / d_c over exactly its neighbours, y_c = b_c) gets
  // x_c = b_c / d_c - sum_i l_ic x_i from the backward sweep, so b_c - K_cc x_c - sum_i K_ic x_i is the rounding of these few
  // operations whatever the other pivots did -- relative to the row's own terms always O(eps).  (C3: 4000 of 9200 rows, 22 %
  // of the records; a refinement step leaves them at zero: the residual vector is cleared when it is allocated.)
  {
    std::vector<int> kept;
    kept.reserve((size_t)n);
    for (int c : order) {
      const int pv = P.piv_of_col[(size_t)c];
      const bool exact = P.piv_w[(size_t)pv] == 1 && P.sfwd_eptr[(size_t)c + 1] == P.sfwd_eptr[(size_t)c] &&
                         (P.piv_chain.empty() || P.piv_chain[(size_t)pv] < 0) && pv != P.front_piv;
      if (!exact) kept.push_back(c);
    }
    order.swap(kept);
  }
  const int nrows = (int)order.size();
  g->res_nrows = nrows;
  std::vector<int> ptr((size_t)nrows + 1, 0), vraw, xnew, xold, brow_new((size_t)std::max(nrows, 1)), brow_old((size_t)std::max(nrows, 1));
  for (int pos = 0; pos < nrows; ++pos) {
    const int c = order[(size_t)pos];
    brow_new[(size_t)pos] = c;
    brow_old[(size_t)pos] = P.perm[(size_t)c];
    for (auto& rc : rows[(size_t)c])
      for (int q = g->can_ptr[(size_t)rc[0]]; q < g->can_ptr[(size_t)rc[0] + 1]; ++q) {
        vraw.push_back(rawmap[(size_t)g->can_idx[(size_t)q]]);
        xnew.push_back(rc[1]);
        xold.push_back(rc[1] >= 0 ? P.perm[(size_t)rc[1]] : rc[1]);
      }
    ptr[(size_t)pos + 1] = (int)vraw.size();
  }
  if (int rcb = dev_upload(h, g, &g->res_brow_new, brow_new)) return rcb;
  if (int rcb = dev_upload(h, g, &g->res_brow_old, brow_old)) return rcb;
  g->res_ne = (int)vraw.size();
  g->res_vraw_host = vraw;
  {
    // the border by (local) coupling row: sum_e A[c, j] x_j
    const int ncl = g->nc_loc;
    std::vector<std::vector<std::array<int, 2>>> brows((size_t)std::max(ncl, 1));
    for (int e = 0; e < nnzB; ++e) brows[(size_t)rowB[e]].push_back({nnzK + e, P.iperm[(size_t)colB[e]]});
    std::vector<int> bptr((size_t)ncl + 1, 0), bv, bxn, bxo;
    for (int c = 0; c < ncl; ++c) {
      for (auto& rc2 : brows[(size_t)c])
        for (int q = g->can_ptr[(size_t)rc2[0]]; q < g->can_ptr[(size_t)rc2[0] + 1]; ++q) {
          bv.push_back(rawmap[(size_t)g->can_idx[(size_t)q]]);
          bxn.push_back(rc2[1]);
          bxo.push_back(P.perm[(size_t)rc2[1]]);
        }
      bptr[(size_t)c + 1] = (int)bv.size();
    }
    g->res_bne = (int)bv.size();
    g->res_bvraw_host = bv;
    int rcb;
    if ((rcb = dev_upload(h, g, &g->res_bptr, bptr))) return rcb;
    if ((rcb = dev_upload(h, g, &g->res_bvraw, bv))) return rcb;
    if ((rcb = dev_upload(h, g, &g->res_bxnew, bxn))) return rcb;
    if ((rcb = dev_upload(h, g, &g->res_bxold, bxo))) return rcb;
    if ((rcb = dev_alloc(h, g, &g->res_bpart, (size_t)2 * std::max(ncl, 1) * (size_t)g->dev.nchunk))) return rcb;
  }
  for (int q = 0; q < 4; ++q) { vraw.push_back(0); xnew.push_back(0); xold.push_back(0); }
  int rc;
  if ((rc = dev_upload(h, g, &g->res_ptr, ptr))) return rc;
  if ((rc = dev_upload(h, g, &g->res_vraw, vraw))) return rc;
  if ((rc = dev_upload(h, g, &g->res_xnew, xnew))) return rc;
  if ((rc = dev_upload(h, g, &g->res_xold, xold))) return rc;
  if ((rc = dev_alloc(h, g, &g->res_rmax, (size_t)g->dev.bpad))) return rc;
  if ((rc = dev_alloc(h, g, &g->res_smax, (size_t)g->dev.bpad))) return rc;
  PP_HIP(hipMemset(g->res_rmax, 0, (size_t)g->dev.bpad * sizeof(unsigned long long)));
  PP_HIP(hipMemset(g->res_smax, 0, (size_t)g->dev.bpad * sizeof(unsigned long long)));
  return 0;
}

// The same records over the producer's source rows (api.hip: pp_set_value_map; ms / mc: source row or -1 and
// coefficient of every compact row of the transposed input).
int ppi_residual_value_map(pp_handle h, Group* g, const std::vector<int>& ms, const std::vector<double>& mc) {
  std::vector<int> vs((size_t)g->res_ne + 4, 0);
  std::vector<double> cs((size_t)g->res_ne + 4, 0.0);
  for (int e = 0; e < g->res_ne; ++e) {
    const int row = g->res_vraw_host[(size_t)e];
    vs[(size_t)e] = ms[(size_t)row];        // (-1: the constant)
    cs[(size_t)e] = mc[(size_t)row];
  }
  std::vector<int> bvs((size_t)g->res_bne + 1, 0);
  std::vector<double> bcs((size_t)g->res_bne + 1, 0.0);
  for (int e = 0; e < g->res_bne; ++e) {
    const int row = g->res_bvraw_host[(size_t)e];
    bvs[(size_t)e] = ms[(size_t)row];
    bcs[(size_t)e] = mc[(size_t)row];
  }
  for (void* p : {(void*)g->res_vsrc, (void*)g->res_csrc, (void*)g->res_bvsrc, (void*)g->res_bcsrc}) if (p) (void)hipFree(p);
  g->res_vsrc = g->res_bvsrc = nullptr; g->res_csrc = g->res_bcsrc = nullptr;
  int rc;
  if ((rc = dev_alloc(h, (Group*)nullptr, &g->res_vsrc, vs.size()))) return rc;
  if ((rc = dev_alloc(h, (Group*)nullptr, &g->res_csrc, cs.size()))) return rc;
  if ((rc = dev_alloc(h, (Group*)nullptr, &g->res_bvsrc, bvs.size()))) return rc;
  if ((rc = dev_alloc(h, (Group*)nullptr, &g->res_bcsrc, bcs.size()))) return rc;
  PP_HIP(hipMemcpy(g->res_vsrc, vs.data(), vs.size() * sizeof(int), hipMemcpyHostToDevice));
  PP_HIP(hipMemcpy(g->res_csrc, cs.data(), cs.size() * sizeof(double), hipMemcpyHostToDevice));
  PP_HIP(hipMemcpy(g->res_bvsrc, bvs.data(), bvs.size() * sizeof(int), hipMemcpyHostToDevice));
  PP_HIP(hipMemcpy(g->res_bcsrc, bcs.data(), bcs.size() * sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

extern "C" {

int pp_residual(pp_handle h, int store, const double* bc_dev, int coupling_on_device) {
  if (!h || !h->schur_done) return fail(h, 3, "pp_residual before a back-solve");
  PP_HIP(hipSetDevice(h->device));
  const int nc = h->nc;
  const size_t ncp = (size_t)std::max(nc, 1);
  if (!h->resid_host) {
    // (the device buffers first: the mailbox pointer is what says "allocated", so a failure half-way is retried as a whole)
    if (!h->resid_best) { if (int rc = dev_alloc<double>(h, nullptr, &h->resid_best, 4)) return rc; }
    if (!h->resid_ax) { if (int rc = dev_alloc<double>(h, nullptr, &h->resid_ax, 2 * ncp + 2 * 1024)) return rc; }      // (+ a slot pair per rank)
    if (!h->resid_rc) { if (int rc = dev_alloc<double>(h, nullptr, &h->resid_rc, 2 * ncp)) return rc; }         // r_c | its row scales
    void* hp = nullptr;
    void* dp = nullptr;
    const size_t doubles = 8 + 4 * ncp;
    PP_HIP(hipHostMalloc(&hp, doubles * sizeof(double), hipHostMallocMapped));
    std::memset(hp, 0, doubles * sizeof(double));
    if (hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) { (void)hipHostFree(hp); return fail(h, 3, "hipHostGetDevicePointer failed (check mailbox)"); }
    h->resid_dev = (double*)dp;
    h->resid_host = (volatile double*)hp;
  }
  if (int rc = join_dense(h)) return rc;
  if (!bc_dev) bc_dev = h->last_rc;
  const hipStream_t st = h->stream;
  const size_t ng = h->groups.size();
  ++h->resid_seq;
  // coupling rows on the device: a dense S whose sums are complete on this rank (the caller says so); Q as pp_factor_schur got it
  // coupling_on_device == 2: several ranks whose sums meet through the library's communicator (one all-reduce, below)
  const bool ranks = coupling_on_device == 2 && h->rccl_comm && h->rccl_ranks >= 1 && h->rccl_ranks <= 1024;
  if (coupling_on_device == 2 && !ranks) return fail(h, 3, "pp_residual: no communicator for the check across ranks (pp_comm_init)");
  // Q: dense (1) / none (0) as the last pp_factor_schur got it; block-tridiagonal S: the pairs of pp_factor_schur_corner (2);
  // a block-tridiagonal S factorised from a flat Q leaves the coupling rows to the caller
  int Qfin = -1;
  if (coupling_on_device) Qfin = h->btd ? (h->corner_nnz >= 0 ? 2 : -1) : (h->have_Q ? 1 : 0);
  // (a block-tridiagonal S factorised from a flat Q: one rank hands the coupling rows to the caller, as with coupling_on_device = 0)
  if (ranks && Qfin < 0) return fail(h, 3, "pp_residual: no Q on the device for the coupling rows (block-tridiagonal S factorised from a flat Q)");
  const int Qmode = (ranks || Qfin == 2) ? -2 : Qfin;
  h->resid_rc_valid = false;
  bool any_mapped = ng == 0;
  for (Group* g : h->groups) any_mapped = any_mapped || (g->dev.cmapT != nullptr) || g->dev.nc != nc;
  if (any_mapped) PP_HIP(hipMemsetAsync(h->resid_ax, 0, 2 * ncp * sizeof(double), st));
  static const int rows_env = pp::env_switch("PP_RES_ROWS") ? std::atoi(pp::env_switch("PP_RES_ROWS")) : 0;
  if (ng == 0) {                 // no local block: x_c and b_c only, an empty (passing) block result
    hipLaunchKernelGGL(k_residual_reduce, dim3(1), dim3(256), 0, st, (unsigned long long*)nullptr, (unsigned long long*)nullptr, 0, -1, 1, 1,
                       h->resid_best, h->resid_dev, h->resid_seq, nc, 0, 0, (const double*)nullptr, h->resid_ax, (const double*)h->xc, bc_dev,
                       Qmode, (const double*)h->Qd, store ? h->resid_rc : (double*)nullptr);
  }
  for (size_t gi = 0; gi < ng; ++gi) {
    Group* g = h->groups[gi];
    GroupDev d = g->dev;
    const pp::Plan& P = g->plan;
    const bool native = g->x_native != nullptr && g->rhs_native != nullptr;
    if (!native && (!d.X || !d.rhs)) return fail(h, 3, "pp_residual: no right-hand side / solution of the last back-solve");
    ResArgs a{};
    a.rptr = g->res_ptr;
    a.bptr = g->res_bptr;
    if (g->last_fused) {
      if (!g->res_vsrc || !g->src) return fail(h, 3, "pp_residual: no value map / source buffer");
      a.V = g->src; a.vrow = g->res_vsrc; a.coef = g->res_csrc; a.bvrow = g->res_bvsrc; a.bcoef = g->res_bcsrc;
    } else {
      if (!d.rawT) return fail(h, 3, "pp_residual: the values of the last factorisation are gone");
      a.V = d.rawT; a.vrow = g->res_vraw; a.bvrow = g->res_bvraw;
    }
    if (native) { a.B = g->rhs_native; a.X = g->x_native; a.xcol = g->res_xold; a.brow = g->res_brow_old; a.bxcol = g->res_bxold; }
    else {
      // b was consumed by the forward sweep (y is computed in place): transposed into Y again, elimination order
      const int tiles = transpose_tiles(P.n, d.nchunk);
      hipLaunchKernelGGL(k_transpose_in, dim3((unsigned)((P.n + 64 * tiles - 1) / (64 * tiles)) * d.nchunk), dim3(256), 0, st, d.rhs,
                         d.Y, d.iperm, d.batch, P.n, d.bpad, tiles, (const int*)nullptr);
      a.B = d.Y; a.X = d.X; a.xcol = g->res_xnew; a.brow = g->res_brow_new; a.bxcol = g->res_bxnew;
    }
    if (store) {
      if (!g->res_R) {
        std::lock_guard<std::mutex> lk(h->alloc_mu);
        if (int rc = value_alloc(h, g, &g->res_R, (size_t)P.n * (size_t)d.bpad)) return rc;
        if (int rc = value_alloc(h, g, &g->res_D, (size_t)P.n * (size_t)d.bpad)) return rc;
        PP_HIP(hipMemsetAsync(g->res_R, 0, (size_t)P.n * (size_t)d.bpad * sizeof(double), st));      // (rows that are not evaluated stay zero)
      }
      a.Rout = g->res_R;             // caller's row order: the correction solve runs on native vectors
      a.rrow = g->res_brow_old;
    }
    a.xc = (d.cmapT && d.nc > 0) ? d.XCL : h->xc;
    a.rmax = g->res_rmax; a.smax = g->res_smax;
    a.bpart = g->res_bpart; a.ax = h->resid_ax; a.nc_glob = nc;
    // rows per wave: enough waves to fill the chip several times over, few enough atomics (measured at C3: tools/sweep_env.sh PP_RES_ROWS)
    // (x RES_NW = 8 waves: 8 rows per workgroup.  MEASURED at C3, kernel alone: 1 wave x 8 rows 97 us; reverse Cuthill-McKee order
    // 4 x 8 rows 83.5; 8 x 2 rows 69.1; 16 x 2 rows 78.1; 16 x 1 94.9; without the rows that hold by construction 8 x 1 / 2 / 3 / 4
    // rows: 51.6 / 56.2 / 55.0 / 57.7.  With few instances the launch is not bound by bandwidth but by the chain of atomic maxima on
    // each instance's word (one per workgroup and instance): 128 blocks 31.6 / 21.5 / 17.3 / 18.8 / 34.2 us for 1 / 2 / 4 / 8 / 16 rows
    // per wave, 256 blocks 32.4 / 25.2 / 20.9 / 21.9 / 37.0, 512 blocks 37.7 / 35.0 / 32.9 / 36.2 / 41.0: four rows up to 8 chunks)
    a.rows_per_wg = rows_env > 0 ? rows_env : (d.nchunk <= 8 ? 4 : 1);
    a.nrows = g->res_nrows;
    const unsigned ntask = (unsigned)((g->res_nrows + a.rows_per_wg * RES_NW - 1) / (a.rows_per_wg * RES_NW));
    const unsigned nborder = ((unsigned)d.nc * (unsigned)d.nchunk + RES_NW - 1) / RES_NW;
    const bool pair = h->lane_pairs && d.nchunk % 2 == 0;
    a.ny = pair ? d.nchunk / 2 : d.nchunk;
    a.nres_wg = (int)(ntask * (unsigned)a.ny);
#define PP_LAUNCH_RES(NV, ST) hipLaunchKernelGGL((k_residual<NV, ST>), dim3((unsigned)a.nres_wg + nborder), dim3(64 * RES_NW), 0, st, d, a, a.rptr, \
                                                a.vrow, a.xcol, a.brow, a.coef, a.V, a.B, a.X, a.xc)
    if ((unsigned)a.nres_wg + nborder == 0u) {}          // (every row holds by construction and there is no coupling row: nothing to launch)
    else if (pair) { if (store) PP_LAUNCH_RES(2, true); else PP_LAUNCH_RES(2, false); }
    else { if (store) PP_LAUNCH_RES(1, true); else PP_LAUNCH_RES(1, false); }
#undef PP_LAUNCH_RES
    const bool uniform = !d.cmapT && d.nc == nc && d.nc > 0;
    hipLaunchKernelGGL(k_residual_reduce, dim3(1), dim3(256), 0, st, g->res_rmax, g->res_smax, d.batch, (int)gi, gi == 0 ? 1 : 0,
                       gi + 1 == ng ? 1 : 0, h->resid_best, h->resid_dev, h->resid_seq, nc, d.nc, d.nchunk,
                       uniform ? (const double*)g->res_bpart : (const double*)nullptr, h->resid_ax, (const double*)h->xc, bc_dev, Qmode,
                       (const double*)h->Qd, store ? h->resid_rc : (double*)nullptr);
  }
  if (ranks || Qfin == 2) {
    if (ranks) {
      hipLaunchKernelGGL(k_residual_slots, dim3(1), dim3(64), 0, st, (const double*)h->resid_best, h->resid_ax, nc, h->rccl_ranks, h->rccl_rank);
      if (int rc = ppi_allreduce_sum(h, h->resid_ax, 2 * (size_t)nc + 2 * (size_t)h->rccl_ranks)) return rc;
    }
    if (Qfin == 2 && nc > 0) {
      hipLaunchKernelGGL(k_corner_rc_init, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, st, nc, bc_dev, (const double*)h->resid_ax,
                         h->resid_rc, h->resid_rc + nc);
      if (h->corner_nnz > 0)
        hipLaunchKernelGGL(k_corner_rc_apply, dim3((unsigned)((h->corner_nnz + 255) / 256)), dim3(256), 0, st, (long long)h->corner_nnz,
                           (const long long*)h->corner_pos, (const double*)h->corner_val, h->gs, h->G, (const double*)h->xc, h->resid_rc,
                           h->resid_rc + nc);
    }
    hipLaunchKernelGGL(k_residual_finish, dim3(1), dim3(256), 0, st, (const double*)h->resid_best, h->resid_dev, h->resid_seq, nc,
                       ranks ? h->rccl_ranks : 0, (const double*)h->resid_ax, (const double*)h->xc, bc_dev, Qfin, (const double*)h->Qd,
                       (store || Qfin == 2) ? h->resid_rc : (double*)nullptr, (const double*)(h->resid_rc + nc));
    if (store) h->resid_rc_valid = true;
  } else if (Qmode < 0) {
    // the caller finishes the coupling rows: x_c | sum A x | sum |A||x| | b_c behind the header, then the flag
    if (nc > 0) {
      double* mail = const_cast<double*>(h->resid_host);
      PP_HIP(hipMemcpyAsync(mail + 8, h->xc, (size_t)nc * sizeof(double), hipMemcpyDeviceToHost, st));
      PP_HIP(hipMemcpyAsync(mail + 8 + nc, h->resid_ax, 2 * (size_t)nc * sizeof(double), hipMemcpyDeviceToHost, st));
      if (bc_dev) PP_HIP(hipMemcpyAsync(mail + 8 + 3 * (size_t)nc, bc_dev, (size_t)nc * sizeof(double), hipMemcpyDeviceToHost, st));
      else std::memset(mail + 8 + 3 * (size_t)nc, 0, (size_t)nc * sizeof(double));
    }
    hipLaunchKernelGGL(k_residual_flag, dim3(1), dim3(1), 0, st, h->resid_dev, h->resid_seq);
  } else if (store) {
    h->resid_rc_valid = true;
  }
  h->resid_on_device = Qfin >= 0;
  PP_HIP(hipGetLastError());
  return 0;
}

int pp_residual_result(pp_handle h, double out[6], double* coupling_out) {
  if (!h || !h->resid_host || h->resid_seq == 0) return fail(h, 3, "pp_residual_result before pp_residual");
  PP_HIP(hipSetDevice(h->device));
  const volatile long long* seq = reinterpret_cast<volatile long long*>(h->resid_host) + 3;
  const auto t0 = std::chrono::steady_clock::now();
  int spins = 0;
  while (__atomic_load_n(const_cast<const long long*>(seq), __ATOMIC_ACQUIRE) != h->resid_seq) {
    if (++spins % 4096 == 0) {
      if (hipStreamQuery(h->stream) == hipSuccess && __atomic_load_n(const_cast<const long long*>(seq), __ATOMIC_ACQUIRE) != h->resid_seq) {
        // the stream has drained and the mailbox is still stale: a launch failed
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) return fail(h, 3, "pp_residual_result: the check did not run");
      }
    }
  }
  out[0] = h->resid_host[0]; out[1] = h->resid_host[1]; out[2] = h->resid_host[2]; out[3] = h->resid_host[4];
  out[4] = h->resid_on_device ? h->resid_host[5] : -1.0;      // rho of the coupling rows, or -1: the caller finishes them
  out[5] = h->resid_on_device ? h->resid_host[6] : h->resid_host[0];      // worst block of ANY rank (coupling_on_device == 2), else this rank's
  if (coupling_out && !h->resid_on_device)       // x_c | sum A x | sum |A||x| | b_c, n_c doubles each
    for (size_t i = 0; i < 4 * (size_t)h->nc; ++i) coupling_out[i] = h->resid_host[8 + i];
  return 0;
}

// The coupling solve of a correction solve whose right-hand side -- the residual of the coupling rows -- pp_residual(h, 1, ., 1)
// left on the device.
int pp_refine_solve_coupling(pp_handle h) {
  if (!h || !h->refining || !h->resid_rc_valid) return fail(h, 3, "pp_refine_solve_coupling: no residual of the coupling rows on the device");
  return pp_solve_coupling_dev(h, h->nc > 0 ? h->resid_rc : nullptr);
}

// Correction solve of a refinement step: between begin and end the sweeps (pp_solve_forward, the caller's all-reduce of
// r_s, pp_solve_coupling_dev(NULL), pp_solve_backward) run on the stored residual (pp_residual(h, 1)) as right-hand side
// and write the correction d; end adds d to the solution of the back-solve (and its coupling part to x_c) and restores
// the caller's vectors.
int pp_refine_begin(pp_handle h) {
  if (!h || !h->schur_done) return fail(h, 3, "pp_refine_begin before a back-solve");
  if (h->refining) return fail(h, 3, "pp_refine_begin: a correction solve is already open");
  PP_HIP(hipSetDevice(h->device));
  for (Group* g : h->groups)
    if (!g->res_R || !g->res_D) return fail(h, 3, "pp_refine_begin: no stored residual (pp_residual(h, 1) first)");
  if (h->nc > 0) {
    if (!h->xc_save) { if (int rc = dev_alloc<double>(h, nullptr, &h->xc_save, (size_t)h->nc)) return rc; }
    PP_HIP(hipMemcpyAsync(h->xc_save, h->xc, (size_t)h->nc * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  }
  for (Group* g : h->groups) {
    g->save_rhs_native = g->rhs_native;
    g->save_x_native = g->x_native;
    g->rhs_native = g->res_R;
    g->x_native = g->res_D;
  }
  h->refining = true;
  return 0;
}

int pp_refine_end(pp_handle h) {
  if (!h || !h->refining) return fail(h, 3, "pp_refine_end without pp_refine_begin");
  PP_HIP(hipSetDevice(h->device));
  const hipStream_t st = h->stream;
  h->refining = false;
  if (h->nc > 0) {
    hipLaunchKernelGGL(k_add_vec, dim3((unsigned)((h->nc + 255) / 256)), dim3(256), 0, st, h->xc, h->xc_save, h->nc);
  }
  for (Group* g : h->groups) {
    GroupDev& d = g->dev;
    const pp::Plan& P = g->plan;
    g->rhs_native = g->save_rhs_native;
    g->x_native = g->save_x_native;
    const bool native = g->x_native != nullptr;
    const unsigned nb = (unsigned)(((size_t)P.n * d.bpad + 255) / 256);
    if (native) {
      hipLaunchKernelGGL(k_add_rows, dim3(nb), dim3(256), 0, st, g->x_native, (const int*)nullptr, (const double*)g->res_D, (const int*)nullptr,
                         P.n, d.bpad);
    } else {
      // x lives in X in elimination order; d came back in the caller's row order
      hipLaunchKernelGGL(k_add_rows, dim3(nb), dim3(256), 0, st, d.X, (const int*)nullptr, (const double*)g->res_D, d.perm, P.n, d.bpad);
      hipLaunchKernelGGL(k_transpose_out, dim3((unsigned)((P.n + 63) / 64) * d.nchunk), dim3(256), 0, st, d.X, d.iperm, d.xout,
                         d.batch, P.n, d.bpad);
    }
    if (d.cmapT && d.nc > 0)
      hipLaunchKernelGGL(k_gather_xc_refine, dim3((unsigned)(((size_t)d.nc * d.bpad + 255) / 256)), dim3(256), 0, st, d, h->xc);
  }
  PP_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
