// Block factorisation: leaf assembly fused into the left-looking LDL^T in L form (plan.hpp), static block pivots, root front.
// (reference: MA27B numeric factorisation, parapint/linalg/ma27_interface.py:94-141; mpi_explicit_schur_complement.py:292-299)
#include "common.hpp"
#include "kernels_transpose.hpp"

namespace {

template <int WM>
__device__ __forceinline__ void invert_and_scale(const GroupDev& g, int p, int w, int uoff, int doff, unsigned sub,
                                                 int r0, int r1, const double (&tmd)[WM], bool publish, size_t bpad, int b,
                                                 double eps) {
  const double* Up = g.U + (size_t)uoff * bpad + b;
  double* Lp = g.L + (size_t)uoff * bpad + b;
  double inv[WM * (WM + 1) / 2];
  int code;
  if (WM == 1) {
    const pp::PivotResult pr = pp::invert_pivot(1, Up[0], 0.0, 0.0, tmd[0], eps);
    inv[0] = pr.i00;
    code = (pr.code & 3) | (((pr.code >> 2) & 3) << 4) | (((pr.code >> 4) & 3) << 8);
  } else {
    double blk[WM * WM];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WM; ++j)
        blk[i * WM + j] = (i < w && j < w) ? Up[(size_t)(i * w + j) * bpad] : 0.0;
    code = pp::invert_block_t<WM>(w, sub, blk, tmd, eps, inv);
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

template <int WM, int NW, int NV>
__global__ __launch_bounds__(64 * NW) void k_gather_flat(GroupDev g, int task0, int chunk0, int ny, double eps) {
  __shared__ double red[NW > 1 ? NW : 1][NW > 1 ? 2 * WM * NV : 1][NW > 1 ? 64 : 1];   // partial sums / term magnitudes of a split row
  const int lane = threadIdx.x & 63, wave = (NW > 1) ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
  const unsigned b = (unsigned)((((NV == 2 ? PP_PAIR_OF_WG(ny) : PP_CHUNK_OF_WG(ny)) + chunk0) * 64 + lane) * NV);    // first instance of this lane
  const size_t bpad = (size_t)g.bpad;
  const int* t = g.ftask + TASK_INTS * (size_t)(task0 + NW * PP_TASK_OF_WG(ny) + wave);
  const int p = t[0], r0 = t[1], r1 = t[2], kind = t[4], E0 = t[5], E1 = t[6];
  const int piece = (NW > 1) ? t[12] : 0, npieces = (NW > 1) ? t[13] : 1;   // npieces is the same for all waves of the workgroup
  if (kind < 0 && npieces <= 1) return;            // quad padding (in a split quad the padding waves join the barrier)
  const int w = (WM == 1) ? 1 : t[7];
  const int uoff = t[8], boff = t[9], doff = t[10];
  const unsigned sub = (unsigned)t[11];
  const int wp = t[14], qoff = t[15];     // width of the whole panel and first column of this task's slice (root front; else w, 0)
  const double* __restrict__ Ub = g.U;
  const double* __restrict__ Lb = g.L;
  const double* __restrict__ Rb = g.rawT;
  const int nrow = r1 - r0;
  double* Udst = g.U + ((size_t)uoff + (size_t)r0 * wp + qoff) * bpad;     // uniform; lane offset added at the store
  double* Tmd = g.Tm + ((size_t)boff + (size_t)r0 * wp + qoff) * bpad;
  const int nblk = (r0 < wp) ? (wp - r0) : 0;        // leading destination rows that belong to the pivot block
  const bool split = NW > 1 && npieces > 1;
  double tmd[WM][NV];                                 // fused panels: term magnitudes of the diagonal entries of the pivot block
  double acc[WM][NV], tmax[WM][NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
#pragma unroll
    for (int q = 0; q < WM; ++q) { acc[q][v] = 0.0; tmax[q][v] = 0.0; tmd[q][v] = 0.0; }
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
            else if (q == d) {                      // (row d of the block: its diagonal entry is column d)
#pragma unroll
              for (int v = 0; v < NV; ++v) tmd[q][v] = tmax[q][v];
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
          // (column q of a product entry over the columns q0 .. q0 + n - 1 reads row q - q0 of its run; the columns outside
          // the run repeat an end of it and are not used)
          ldv<NV>(Lb + (size_t)(prod ? el[i] + min(max(q - (ef[i] & 15), 0), ((ef[i] >> 4) & 15) - 1) * ew[i] : 0) * bpad + b, sl[i][q]);
      }
#pragma unroll
      for (int i = 0; i < G; ++i) {
        if (i0 + i < cnt) {
          for (int nf = ef[i] >> 8; nf > 0; --nf) finalize();
          if (eu[i] >= 0) {
            // product entry: columns q0 .. q0 + n - 1 of the destination row (all w columns when the source panel holds
            // the whole block pivot, fewer when it holds only some of its columns as rows)
            const unsigned q0 = (unsigned)(ef[i] & 15), qn = (unsigned)((ef[i] >> 4) & 15);
            if (d < nblk) {
#pragma unroll
              for (int q = 0; q < WM; ++q) {
                if ((unsigned)q - q0 < qn) {
#pragma unroll
                  for (int v = 0; v < NV; ++v) {
                    const double term = su[i][v] * sl[i][q][v];
                    acc[q][v] -= term;
                    tmax[q][v] = fmax(tmax[q][v], fabs(term));
                  }
                }
              }
            } else if (q0 == 0u && qn >= (unsigned)w) {
#pragma unroll
              for (int q = 0; q < WM; ++q)     // (q >= w: a duplicate of column w - 1, never stored)
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[q][v] = fma(-su[i][v], sl[i][q][v], acc[q][v]);
            } else {
#pragma unroll
              for (int q = 0; q < WM; ++q) {
                if ((unsigned)q - q0 < qn) {
#pragma unroll
                  for (int v = 0; v < NV; ++v) acc[q][v] = fma(-su[i][v], sl[i][q][v], acc[q][v]);
                }
              }
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
    for (int v = 0; v < NV; ++v) {
      double tv[WM];
#pragma unroll
      for (int q = 0; q < WM; ++q) tv[q] = tmd[q][v];
      invert_and_scale<WM>(g, p, w, uoff, doff, sub, r0, r1, tv, true, bpad, (int)b + v, eps);
    }
  }
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
  double tmd[WM][NV], inv1[NV], lmax = 0.0;           // (tmd: term magnitudes of the diagonal entries of a fused panel's block)
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    inv1[v] = 0.0;
#pragma unroll
    for (int q = 0; q < WM; ++q) tmd[q][v] = 0.0;
  }
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
        const int q0 = ew & 15, qn = (ew >> 4) & 15;      // product entry: columns q0 .. q0 + qn - 1
        ldv<NV>(Lb + (size_t)((ex >= 0) ? ey + min(max(q - q0, 0), qn - 1) * ez : 0) * bpad, lv);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          // (product entry over a run of columns; initial value into column ew & 0xff)
          const double m = (ex >= 0) ? (((unsigned)(q - q0) < (unsigned)qn) ? lv[v] : 0.0) : ((q == (ew & 0xff)) ? -coef : 0.0);
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
          else if (q == d) {
#pragma unroll
            for (int v = 0; v < NV; ++v) tmd[q][v] = tmax[q][v];
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
    for (int v = 0; v < NV; ++v) {
      double tv[WM];
#pragma unroll
      for (int q = 0; q < WM; ++q) tv[q] = tmd[q][v];
      invert_and_scale<WM>(g, p, w, uoff, doff, sub, r0, r1, tv, true, bpad, (int)b + v, eps);
    }
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
  double tmd[WM], blk[WM * WM], u[RV];    // (tmd: term magnitudes of the diagonal entries, pivot.hpp)
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const double tv = Tmp[(size_t)(i < w ? i * w + i : 0) * bpad];
    tmd[i] = (i < w) ? tv : 0.0;
#pragma unroll
    for (int j = 0; j < WM; ++j) {
      const bool in = i < w && j < w;
      const double uv = Up[(size_t)(in ? i * w + j : 0) * bpad];
      blk[i * WM + j] = in ? uv : 0.0;
    }
  }
  const int v0 = r0 * w, v1 = r1 * w;      // value range [v0, v1) of this chunk in the panel
#pragma unroll
  for (int i = 0; i < RV; ++i) u[i] = Up[(size_t)min(v0 + i, v1 - 1) * bpad];
  double inv[WM * (WM + 1) / 2];
  const int code = pp::invert_block_t<WM>(w, sub, blk, tmd, eps, inv);
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
// Tile task (plan.hpp, kind 4): PP_TILE_ROWS consecutive rows of a panel gathered together against whole source panels.
// Device form (api.hip): a stream of STEPS, one per source column: {U position of the tile's row i x 4, L position of the
// panel's column q x 4} (a row or column the source lacks points at the zero rows behind the panels) = 4 + 4 operand loads
// for 16 multiply-adds, against 1 + w loads per w multiply-adds and a 16-byte record per source column of every ROW in the
// row tasks.  The stream is uniform and branch-free (every task padded to a multiple of PP_TILE_DEPTH steps with all-zero
// steps), so the loop is a software pipeline: the operands of step s + PP_TILE_DEPTH - 1 are requested before the
// multiply-adds of step s (MEASURED, C4, the first form of this kernel -- one record per source panel, its 32 loads
// requested, waited for and used: 1.8 us per record at 2 waves per SIMD, every launch a chain of such round trips).
// The rows of a chain front's panels are dense against the fronts below them: that is where these tasks are planned.
// A long tile is cut into pieces by source panels (the waves of one workgroup); partial sums meet in LDS and wave 0 adds
// them in piece order.  Initial values come through ordinary records (fent: the fused-source form is applied to them like
// to all others).  Term magnitudes are kept for the diagonal entries of pivot-block rows (what the zero-pivot test reads):
// a tile that holds pivot rows starts at row 0 of its panel, so the diagonal entry of tile row i is column i.
template <bool PIVT>
__device__ __forceinline__ void tile_stream(const int* __restrict__ trec, int s0, int s1, const double* __restrict__ Ub,
                                            const double* __restrict__ Lb, unsigned b, double (&acc)[PP_TILE_ROWS][PP_WMAX],
                                            double (&tm)[PP_TILE_ROWS]) {
  constexpr int TR = PP_TILE_ROWS, WM = PP_WMAX, D = PP_TILE_DEPTH;
  // a step's eight operand rows are given in units of 64 doubles (position x chunks of the group: api.hip), so an address
  // is base + (row << 6): no multiplication; the records of D steps are read together (two scalar loads)
  double u[D][TR], l[D][WM];
  const unsigned lane_bytes = b * 8u;
  // (uniform row base + 32-bit lane offset: the loads take their base from scalar registers, no 64-bit vector address each)
  auto request = [&](int slot, const int (&rp)[TR + WM]) {
#pragma unroll
    for (int i = 0; i < TR; ++i)
      u[slot][i] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(Ub + ((size_t)(unsigned)rp[i] << 6)) + lane_bytes);
#pragma unroll
    for (int q = 0; q < WM; ++q)
      l[slot][q] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(Lb + ((size_t)(unsigned)rp[TR + q] << 6)) + lane_bytes);
  };
  {
    int rec[D - 1][TR + WM];
#pragma unroll
    for (int j = 0; j < D - 1; ++j)
#pragma unroll
      for (int q = 0; q < TR + WM; ++q) rec[j][q] = trec[(TR + WM) * (size_t)(s0 + j) + q];
#pragma unroll
    for (int j = 0; j < D - 1; ++j) request(j, rec[j]);
  }
  for (int s = s0; s < s1; s += D) {
    int rec[D][TR + WM];                        // steps s + D - 1 .. s + 2 D - 2
#pragma unroll
    for (int j = 0; j < D; ++j)
#pragma unroll
      for (int q = 0; q < TR + WM; ++q) rec[j][q] = trec[(TR + WM) * (size_t)(s + D - 1 + j) + q];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      request((j + D - 1) % D, rec[j]);
      // (without the barriers the scheduler hoists the requests of all D steps to the top of the iteration and sinks their
      // multiply-adds to its end, behind a wait for ALL loads: the pipeline is drained at every iteration)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TR; ++i) {
#pragma unroll
        for (int q = 0; q < WM; ++q) acc[i][q] = fma(-u[j][i], l[j][q], acc[i][q]);
        if (PIVT) tm[i] = fmax(tm[i], fabs(u[j][i] * l[j][i]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_gather_tiles(GroupDev g, int task0, int ny, double eps) {
  constexpr int TR = PP_TILE_ROWS, WM = PP_WMAX;
  static_assert(TR == WM, "the diagonal entry of tile row i is column i");
  __shared__ double red[NW > 1 ? NW - 1 : 1][TR * WM + TR][64];
  const int lane = threadIdx.x & 63, wave = (NW > 1) ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
  const unsigned b = (unsigned)(PP_CHUNK_OF_WG(ny) * 64 + lane);
  const size_t bpad = (size_t)g.bpad;
  const int* t = g.ttask + TASK_INTS * (size_t)(task0 + NW * PP_TASK_OF_WG(ny) + wave);
  const int r0 = t[1], r1 = t[2], s0 = t[3], kind = t[4], E0 = t[5], E1 = t[6], w = t[7], uoff = t[8], boff = t[9], s1 = t[10];
  const int npieces = (NW > 1) ? t[13] : 1;
  const int wp = t[14], qoff = t[15];
  if (kind < 0 && npieces <= 1) return;            // quad padding (in a split quad the padding waves join the barrier)
  const double* __restrict__ Rb = g.rawT + b;
  double acc[TR][WM], tm[TR];
#pragma unroll
  for (int i = 0; i < TR; ++i) {
    tm[i] = 0.0;
#pragma unroll
    for (int q = 0; q < WM; ++q) acc[i][q] = 0.0;
  }
  const bool pivt = r0 < wp;                       // (uniform) the tile holds rows of the pivot block: r0 == 0, qoff == 0
  if (kind >= 0) {
    // initial values (scalar record reads: a handful per tile)
    int d = 0;
    for (int e = E0; e < E1; ++e) {
      const int* rp = g.fent + 4 * (size_t)e;
      const int ex = rp[0], ey = rp[1], ez = rp[2], ew = rp[3];
      d += ew >> 8;
      const int q = ew & 0xff;
      const bool cst = (-1 - ex) == g.const_row;
      const double coef = __hiloint2double(ez, ey);
      const double v = (cst ? 1.0 : Rb[(size_t)(cst ? 0 : -1 - ex) * bpad]) * coef;
#pragma unroll
      for (int i = 0; i < TR; ++i) {
        if (i == d) {
#pragma unroll
          for (int qq = 0; qq < WM; ++qq)
            if (qq == q) acc[i][qq] += v;
          if (pivt && q == i) tm[i] = fmax(tm[i], fabs(v));
        }
      }
    }
    if (pivt) tile_stream<true>(g.trec, s0, s1, g.U, g.L, b, acc, tm);
    else tile_stream<false>(g.trec, s0, s1, g.U, g.L, b, acc, tm);
  }
  if (NW > 1 && npieces > 1) {
    if (wave > 0) {
#pragma unroll
      for (int i = 0; i < TR; ++i) {
#pragma unroll
        for (int q = 0; q < WM; ++q) red[wave - 1][i * WM + q][lane] = acc[i][q];
        red[wave - 1][TR * WM + i][lane] = tm[i];
      }
    }
    __syncthreads();
    if (wave > 0 || kind < 0) return;
    for (int j = 1; j < npieces; ++j) {
#pragma unroll
      for (int i = 0; i < TR; ++i) {
#pragma unroll
        for (int q = 0; q < WM; ++q) acc[i][q] += red[j - 1][i * WM + q][lane];
        tm[i] = fmax(tm[i], red[j - 1][TR * WM + i][lane]);
      }
    }
  }
  double* Ud = g.U + ((size_t)uoff + (size_t)r0 * wp + qoff) * bpad + b;
#pragma unroll
  for (int i = 0; i < TR; ++i) {
    if (r0 + i < r1) {
#pragma unroll
      for (int q = 0; q < WM; ++q)
        if (q < w) Ud[(size_t)(i * wp + q) * bpad] = acc[i][q];
      if (pivt && i < wp)
        g.Tm[((size_t)boff + (size_t)i * wp + (size_t)i) * bpad + b] = tm[i];
    }
  }
}

// ------------------------------------------------------------------------------------------
// Chain front (plan.hpp, PlanOptions::chain_fronts): a chain of block pivots whose row sets nest exactly -- one dense
// supernode that the <= PP_WMAX-column panels cut into dependent levels (the time blocks of a dynamic problem: fronts of
// 100-170 rows, four columns per level, 10-14 levels each).  The rows of ALL its panels have gathered their contributions
// from outside the chain (ordinary tasks, one launch); here ONE workgroup per instance holds the front's pivot columns in
// LDS and runs the panels one after the other: inversion of the pivot block (pivot.hpp, the same static sequence of 1x1 /
// 2x2 sub-pivots and the same zero-pivot rule as the level kernels), L rows = U rows inv(P), update of the later panels'
// columns  F[r][c] -= sum_k U_p[r][k] L_p[c][k]  (every row and column of a later panel is a row of this one), term
// magnitudes of the later diagonal entries tracked for their zero-pivot tests.  The results go to the same U / L / Dinv /
// codes storage, so the Schur update and the solve sweeps do not know the difference (tests/hostsim mirrors the sums).
// Instances are dealt so that the eight that share a 64-byte sector of every [entry][instance] row run on one XCD at
// about the same time: their strided 8-byte accesses meet in that XCD's L2.
// LDS: F[m][W | 1], Lp[m][4] (scaled rows of the current panel), tmd[W], in doubles.
// (a barrier that orders LDS traffic only: __syncthreads() would also wait for the outstanding global stores of U and L,
// which are never read back here)
__device__ __forceinline__ void chain_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Pivot block of panel (w, sub) at front column c0: inverted by every lane of ONE wave from LDS broadcasts (pivot.hpp: the
// static sub-pivots and zero-pivot rule of the level kernels); lane 0 leaves inv(P) (packed) and the inertia code in LDS
// for the workgroup and writes them to the factor storage.
__device__ __forceinline__ void chain_invert(const GroupDev& g, const double* F, int LD, const double* tmd, int c0, int w, unsigned sub,
                                             int p, int doff, double eps, double* sinv, size_t bpad, int b, int lane) {
  double blk[PP_WMAX * PP_WMAX], tv[PP_WMAX], inv[PP_WMAX * (PP_WMAX + 1) / 2];
#pragma unroll
  for (int a = 0; a < PP_WMAX; ++a) {
    tv[a] = (a < w) ? tmd[c0 + min(a, w - 1)] : 0.0;
#pragma unroll
    for (int c = 0; c < PP_WMAX; ++c) {
      const bool in = a < w && c < w;
      const double v = F[(size_t)(c0 + (in ? a : 0)) * LD + c0 + (in ? c : 0)];
      blk[a * PP_WMAX + c] = in ? v : 0.0;
    }
  }
  const int code = pp::invert_block_t<PP_WMAX>(w, sub, blk, tv, eps, inv);
  if (lane == 0) {
    double* invp = g.Dinv + (size_t)doff * bpad + b;
#pragma unroll
    for (int q = 0; q < PP_WMAX * (PP_WMAX + 1) / 2; ++q) {
      sinv[q] = inv[q];
      if (q < w * (w + 1) / 2) invp[(size_t)q * bpad] = inv[q];
    }
    g.codes[(size_t)p * bpad + b] = (b < g.batch) ? (unsigned short)code : (unsigned short)0;
  }
}

// Round 5, second form: the inversion of the NEXT panel's pivot block leaves the critical path.  After the rows of panel i
// are scaled, wave 0 alone applies panel i to the w x w pivot block of panel i + 1 and inverts it, while the other waves
// apply panel i to everything else; one barrier later every thread reads inv(P_(i+1)) from LDS.  (Per-station stamps of the
// first form, one workgroup, 16 panels: invert 1.2-1.6 us + scale 0.9 + update 1.9-3.1 per panel, all in sequence.)
template <int CHAIN_THREADS>
__global__ __launch_bounds__(CHAIN_THREADS) void k_chain_front(GroupDev g, int front0, double eps) {
  extern __shared__ __attribute__((aligned(16))) double fsh[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned j = blockIdx.x % (unsigned)g.bpad;
  const int fi = front0 + (int)(blockIdx.x / (unsigned)g.bpad);
  const int b = (int)((j & 7u) * ((unsigned)g.bpad >> 3) + (j >> 3));
  const int* H = g.chain_hdr + 8 * (size_t)fi;
  const int m = H[0], W = H[1], npan = H[2];
  const int* PR = g.chain_pan + 8 * (size_t)H[3];
  const int LD = W | 1;
  double* F = fsh;
  double* Lp = F + (size_t)m * LD;
  double* tmd = Lp + 4 * (size_t)m;
  double* sinv = tmd + W;                        // inv(P) of the panel about to be scaled (packed, 10 doubles)
  const size_t bpad = (size_t)g.bpad;
  for (int i = 0; i < npan; ++i) {
    const int* R = PR + 8 * i;
    const int w = R[1], boff = R[3], c0 = R[6], f = R[7];
    const double* Up = g.U + (size_t)R[2] * bpad + b;
    double* Fp = F + (size_t)c0 * LD + c0;
#pragma unroll 4
    for (int idx = tid; idx < f * w; idx += CHAIN_THREADS) {
      const int t = idx / w, q = idx - t * w;
      Fp[t * LD + q] = Up[(size_t)idx * bpad];
    }
    if (tid < w) tmd[c0 + tid] = g.Tm[((size_t)boff + (size_t)(tid * w + tid)) * bpad + b];
  }
  chain_barrier();
  if (wave == 0) chain_invert(g, F, LD, tmd, PR[6], PR[1], (unsigned)PR[5], PR[0], PR[4], eps, sinv, bpad, b, lane);
  chain_barrier();
  bool grow = false;
  for (int i = 0; i < npan; ++i) {
    const int* R = PR + 8 * i;
    const int w = R[1], c0 = R[6], f = R[7];
    double inv[PP_WMAX * (PP_WMAX + 1) / 2];
#pragma unroll
    for (int q = 0; q < PP_WMAX * (PP_WMAX + 1) / 2; ++q) inv[q] = sinv[q];
    {
      double* Ug = g.U + (size_t)R[2] * bpad + b;
      double* Lg = g.L + (size_t)R[2] * bpad + b;
      for (int t = tid; t < f; t += CHAIN_THREADS) {
        const double* fr = F + (size_t)(c0 + t) * LD + c0;
        double u[PP_WMAX];
#pragma unroll
        for (int q = 0; q < PP_WMAX; ++q) u[q] = (q < w) ? fr[min(q, w - 1)] : 0.0;
#pragma unroll
        for (int q = 0; q < PP_WMAX; ++q)
          if (q < w) Ug[(size_t)(t * w + q) * bpad] = u[q];
        if (t >= w) {
#pragma unroll
          for (int t2 = 0; t2 < PP_WMAX; ++t2) {
            if (t2 < w) {
              double v = 0.0;
#pragma unroll
              for (int t1 = 0; t1 < PP_WMAX; ++t1)
                if (t1 < w) v += u[t1] * PP_INV(inv, t1, t2);
              Lg[(size_t)(t * w + t2) * bpad] = v;
              Lp[(size_t)(c0 + t) * 4 + t2] = v;
              grow = grow || fabs(v) > g.lbound;
            }
          }
        }
      }
    }
    if (i + 1 == npan) break;
    chain_barrier();
    const int c1 = c0 + w;
    const int* Rn = R + 8;
    const int wn = Rn[1];                        // next panel: columns [c1, c1 + wn), its pivot block = rows [c1, c1 + wn)
    if (wave == 0) {
      // critical path: panel i applied to the pivot block of panel i + 1 (lane (a, c) one entry), then its inversion
      if (lane < PP_WMAX * PP_WMAX) {
        const int a = lane / PP_WMAX, c = lane % PP_WMAX;
        if (a < wn && c < wn) {
          double* fr = F + (size_t)(c1 + a) * LD;
          double acc = fr[c1 + c], tm = 0.0;
#pragma unroll
          for (int k = 0; k < PP_WMAX; ++k) {
            if (k < w) {
              const double term = fr[c0 + k] * Lp[(size_t)(c1 + c) * 4 + k];
              acc -= term;
              tm = fmax(tm, fabs(term));
            }
          }
          fr[c1 + c] = acc;
          if (a == c) tmd[c1 + a] = fmax(tmd[c1 + a], tm);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (one wave: its LDS operations complete in order)
      chain_invert(g, F, LD, tmd, c1, wn, (unsigned)Rn[5], Rn[0], Rn[4], eps, sinv, bpad, b, lane);
    } else {
      // everything else of the later columns [c1, W) of the rows [c1, m), four columns of one row per thread
      const int nr = m - c1, ncg = (W - c1 + 3) >> 2;
      for (int idx = tid - 64; idx < nr * ncg; idx += CHAIN_THREADS - 64) {
        const int cg = idx / nr, r = c1 + (idx - cg * nr), C0 = c1 + 4 * cg;
        if (r < C0 - (PP_WMAX - 1)) continue;      // (above the first row of the panel that holds column C0: not part of any panel)
        double* fr = F + (size_t)r * LD;
        double acc[4], u[PP_WMAX];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) acc[cc] = fr[min(C0 + cc, W - 1)];
#pragma unroll
        for (int k = 0; k < PP_WMAX; ++k) u[k] = (k < w) ? fr[c0 + min(k, w - 1)] : 0.0;
        double tm = 0.0;
#pragma unroll
        for (int k = 0; k < PP_WMAX; ++k) {
          if (k < w) {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
              const double l = Lp[(size_t)min(C0 + cc, W - 1) * 4 + k];
              const double term = u[k] * l;
              acc[cc] -= term;
              if (C0 + cc == r) tm = fmax(tm, fabs(term));
            }
          }
        }
        const bool nextblk = r < c1 + wn;             // (rows of the next pivot block: its wn columns are wave 0's)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          if (C0 + cc < W && !(nextblk && C0 + cc < c1 + wn)) fr[C0 + cc] = acc[cc];
        if (r >= C0 && r < C0 + 4 && r < W && !nextblk) tmd[r] = fmax(tmd[r], tm);
      }
    }
    chain_barrier();
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
  __shared__ int cnt[NWV][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * 64 + lane;
  const size_t bpad = (size_t)g.bpad;
  const int w = fr.w;
  double A[RW][WF];                       // rows RW wave .. of the block (all columns, both triangles)
  int codes = 0;                          // inertia counts of the pivots this wave owned: pos | neg << 8 | zero << 16
  double m[RW];                           // largest term summed into the diagonal entry of each row so far (pivot.hpp: tmd)
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int i = RW * wave + r;
    m[r] = (i < w) ? g.Tm[((size_t)fr.boff + (size_t)(i * w + i)) * bpad + b] : 0.0;
#pragma unroll
    for (int j = 0; j < WF; ++j) {
      const bool in = i < w && j < w;
      const int hi = i > j ? i : j, lo = i > j ? j : i;       // (the lower triangle of the gathered block is the matrix)
      const size_t off = (size_t)(in ? hi * w + lo : 0) * bpad + b;
      const double av = g.U[(size_t)fr.uoff * bpad + off];
      A[r][j] = in ? av : 0.0;
    }
  }
  // The step loop is unrolled: every register index below is a constant (a rolled loop selects the pivot column with
  // compare / select pairs per element: 1.75 us per step, measured).  One barrier per 1x1 step: the buffers of a
  // step are written again two EXECUTED steps later, which every wave reaches only through the barrier of the step in
  // between.  The buffer parity therefore follows the executed steps (a uniform counter), not k: a 2x2 step skips k + 1,
  // and k & 1 would hand the step at k + 2 the buffers other waves may still be reading.
  bool second = false;
  int par = 0;
#pragma unroll
  for (int k = 0; k < PP_FRONT_MAX; ++k) {
    if (k >= w) continue;                 // (uniform)
    if (second) { second = false; continue; }
    par ^= 1;
    const int k1 = (k + 1 < WF) ? k + 1 : k;
    const int ow = k / RW, orow = k % RW, ow1 = k1 / RW, orow1 = k1 % RW;     // owners of the pivot rows (constants)
    const bool two = ((fr.sub >> k) & 1u) && (k + 1 < w);
    if (!two) {
      if (wave == ow) {
#pragma unroll
        for (int j = 0; j < WF; ++j) rowk[par][0][j][lane] = A[orow][j];
        const pp::PivotResult pr = pp::invert_pivot(1, A[orow][k], 0.0, 0.0, m[orow], eps);
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
        // (the update of the row's own diagonal entry counts as a term of that sum: column RW wave + r of the pivot row)
        m[r] = fmax(m[r], fabs(l * rowk[par][0][min(RW * wave + r, WF - 1)][lane]));
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
        const pp::PivotResult pr = pp::invert_pivot(2, A[orow][k], rowk[par][1][k][lane], rowk[par][1][k1][lane], 0.0, eps);
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
        {
          const int ci = min(RW * wave + r, WF - 1);
          m[r] = fmax(m[r], fmax(fabs(l0 * rowk[par][0][ci][lane]), fabs(l1 * rowk[par][1][ci][lane])));
        }
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
  for (int r = r0; r < r1; ++r) {
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
}


}  // namespace

extern "C" {

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
    g->last_fused = fused_sources;      // (where the a-posteriori check of the back-solves finds the values: refine.hip)
    {
      int nlaunch = 0;
      for (int l = 0; l < P.n_levels; ++l)
        nlaunch += (P.flevel_ptr[l + 1] > P.flevel_ptr[l]) + (P.slevel_ptr[l + 1] > P.slevel_ptr[l]);
      if (P.front_piv >= 0) nlaunch += 1 + (P.wtasks.empty() ? 0 : 1);
      for (int l = 0; l < P.n_levels; ++l) nlaunch += (P.chain_lvl_ptr[l + 1] > P.chain_lvl_ptr[l]) + (P.tlevel_ptr[l + 1] > P.tlevel_ptr[l]);
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
        if (P.tlevel_ptr[l + 1] > P.tlevel_ptr[l]) {
          // tile tasks of this level (the rows of its chain fronts), quads of tasks
          if (sp.n != 1) return fail(h, 3, "instance splits are not supported together with tile tasks");
          const int tt0 = P.tlevel_ptr[l], ntq = (P.tlevel_ptr[l + 1] - tt0) / PP_QUAD;
          hipLaunchKernelGGL((k_gather_tiles<PP_QUAD>), dim3((unsigned)ntq * (unsigned)d.nchunk), dim3(64 * PP_QUAD), 0, fan[0], d, tt0,
                             d.nchunk, PIVOT_EPS);
        }
        if (P.chain_lvl_ptr[l + 1] > P.chain_lvl_ptr[l]) {
          // chain fronts of this level: one workgroup per (front, instance)
          if (sp.n != 1) return fail(h, 3, "instance splits are not supported together with chain fronts");
          const size_t lds = g->chain_lds[(size_t)l];
          if (lds > 64 * 1024 && !h->chain_lds_attr) {
            std::lock_guard<std::mutex> lk(h->alloc_mu);
            if (!h->chain_lds_attr) {
              if (hipFuncSetAttribute((const void*)k_chain_front<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                  hipFuncSetAttribute((const void*)k_chain_front<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return fail(h, 3, "hipFuncSetAttribute failed (chain fronts)");
              h->chain_lds_attr = true;
            }
          }
          const int nfr = P.chain_lvl_ptr[l + 1] - P.chain_lvl_ptr[l];
          // (MEASURED at C4: one front of 512 instances 105 -> 59 us with 512 threads per workgroup, six fronts 188 -> 286 us --
          // half as many workgroups are resident: the wider workgroup where the launch does not fill the chip anyway)
          if ((size_t)nfr * (size_t)d.bpad <= 512)
            hipLaunchKernelGGL(k_chain_front<512>, dim3((unsigned)nfr * (unsigned)d.bpad), dim3(512), lds, fan[0], d, P.chain_lvl_ptr[l], PIVOT_EPS);
          else
            hipLaunchKernelGGL(k_chain_front<256>, dim3((unsigned)nfr * (unsigned)d.bpad), dim3(256), lds, fan[0], d, P.chain_lvl_ptr[l], PIVOT_EPS);
        }
        if (P.front_piv >= 0 && P.piv_flevel[P.front_piv] == l) {
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
  h->blocks_done_valid = false;
  if (gst.n > 1) {            // what an early forward sweep of the handle's own groups waits for (pp_solve_forward_ex)
    if (!h->ev_blocks_done) PP_HIP(hipEventCreateWithFlags(&h->ev_blocks_done, hipEventDisableTiming));
    PP_HIP(hipEventRecord(h->ev_blocks_done, h->stream));
    h->blocks_done_valid = true;
  }
  PP_HIP(hipGetLastError());
  h->blocks_factored = true;
  h->numeric_done = false;
  h->schur_done = false;
  return 0;
}


}  // extern "C"
