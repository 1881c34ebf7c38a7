// The interior-point step on device-resident iterates (include/parapint_hip.h: pp_ip_*).
// Reference: parapint/interfaces/interface.py:450-465 (barrier diagonals), :496-538 (right-hand side), :562-588 (bound-dual
// steps), schur_complement/sc_ip_interface.py:1683-1696 (link rows, coupling block), mpi_sc_ip_interface.py:470-478
// (all-reduce of the coupling block), algorithms/interior_point.py:174-317 (convergence measures), :619-626 (the step),
// :655-758 (fraction to the boundary).
//
// Layout: every array is [row][instance] with the instance index fastest (the solver's own layout), so a
// wavefront is 64 instances of one row and every access is one coalesced 512-byte request.  Row programs and their term
// lists are wave-uniform (scalar loads).  Reductions are two-stage and deterministic: one partial per workgroup, combined
// in a fixed order by the next kernel; the scalars reach the host through a pinned mailbox (no copy, no stream sync).
#include "common.hpp"

// (the arithmetic of every element follows the reference's formulas operation by operation: no contraction into FMAs, so
// that the tests can compare against a numpy restatement to the last bits)
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ double nmax(double a, double b) { return (a != a || b != b) ? NAN : fmax(a, b); }   // NaN propagates,
__device__ __forceinline__ double nmin(double a, double b) { return (a != a || b != b) ? NAN : fmin(a, b); }   // as numpy's max / min

constexpr int IP_MAXG = 8;        // pattern groups per call
constexpr int IP_STEP_SLOTS = 5;  // partials of k_ip_step: compl(0), compl(mu), max |grad_s L|, sum |bound duals|, sum |duals|
constexpr int IP_ROWS_SLOTS = 3;  // partials of k_ip_rows: primal infeasibility, max |grad_x L|, objective
constexpr int IP_RPW = 4;         // rows per wave in k_ip_rows (at least; more when the launch would exceed IP_ROWS_MAXWG workgroups)
constexpr unsigned IP_EW_MAXWG = 2048;    // workgroups of the elementwise kernels (grid-stride beyond): bounds the partials of
constexpr unsigned IP_ROWS_MAXWG = 16384; // the second reduction stage, which one workgroup per slot combines

__device__ __forceinline__ int ip_nb(const pp_ip_group& g) { return g.n + 2 * g.mi + g.me + g.nfs; }

// workgroup reduction of NV values per thread (256 threads); op 0: nmin, 1: nmax, 2: sum.  Result in red[v][0].
template <int NV>
__device__ __forceinline__ void wg_reduce(double (*red)[256], const double* v, const int* op) {
  for (int q = 0; q < NV; ++q) red[q][threadIdx.x] = v[q];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      for (int q = 0; q < NV; ++q) {
        const double a = red[q][threadIdx.x], b = red[q][threadIdx.x + s];
        red[q][threadIdx.x] = op[q] == 0 ? nmin(a, b) : op[q] == 1 ? nmax(a, b) : a + b;
      }
    }
    __syncthreads();
  }
}

// bounds, bound duals and the W rows of variable row r (r < n: primal, else slack r - n)
struct VarRows { size_t lo, hi, zl, zu; };
__device__ __forceinline__ VarRows var_rows(const pp_ip_group& g, int r) {
  const int nb = ip_nb(g);
  VarRows v;
  if (r < g.n) { v.lo = r; v.hi = g.n + r; v.zl = nb + r; v.zu = nb + g.n + r; }
  else { const int i = r - g.n; v.lo = 2 * g.n + i; v.hi = 2 * g.n + g.mi + i; v.zl = nb + 2 * g.n + i; v.zu = nb + 2 * g.n + g.mi + i; }
  return v;
}

// ---- right-hand side rows of the variables (interface.py:496-525): -(G - mu / (x - l) + mu / (u - x)), G = grad f + J^T y
// for the primals and -y_ineq for the slacks
__global__ __launch_bounds__(256) void k_ip_rhs(pp_ip_group g, double mu) {
  const size_t bpad = (size_t)g.bpad;
  const size_t total = (size_t)(g.n + g.mi) * bpad;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int r = (int)(e / bpad);
    const size_t b = e % bpad;
    const VarRows v = var_rows(g, r);
    const double x = g.W[(size_t)r * bpad + b];
    const double lo = g.bounds[v.lo * bpad + b], hi = g.bounds[v.hi * bpad + b];
    const double grad = r < g.n ? g.G[(size_t)r * bpad + b] : -g.W[(size_t)(g.n + g.mi + g.me + (r - g.n)) * bpad + b];
    g.rhs[(size_t)r * bpad + b] = -((grad - mu / (x - lo)) + mu / (hi - x));
  }
}

// ---- fraction to the boundary (interior_point.py:655-758) with the bound-dual steps of interface.py:562-588 formed on
// the fly: one pass over the primal and the slack rows -> per workgroup {alpha_primal, alpha_dual}
__global__ __launch_bounds__(256) void k_ip_stats(pp_ip_group g, double tau, double mu, double* __restrict__ part, int wg0,
                                                  int nwg) {
  __shared__ double red[2][256];
  const size_t bpad = (size_t)g.bpad;
  const size_t total = (size_t)(g.n + g.mi) * bpad;
  double v[2] = {1.0, 1.0};
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int r = (int)(e / bpad);
    const int b = (int)(e % bpad);
    if (b >= g.batch) continue;
    const VarRows q = var_rows(g, r);
    const double x = g.W[(size_t)r * bpad + b], dx = g.delta[(size_t)r * bpad + b];
    const double lo = g.bounds[q.lo * bpad + b], hi = g.bounds[q.hi * bpad + b];
    const double zl = g.W[q.zl * bpad + b], zu = g.W[q.zu * bpad + b];
    if (dx != dx || x != x) v[0] = NAN;
    if (dx < 0.0 && lo > -INFINITY) v[0] = nmin(v[0], -tau * (x - lo) / dx);
    if (dx > 0.0 && hi < INFINITY) v[0] = nmin(v[0], tau * (hi - x) / dx);
    const double dzl = (mu - zl * dx) / (x - lo) - zl;
    const double dzu = (mu + zu * dx) / (hi - x) - zu;
    if (dzl != dzl || zl != zl || dzu != dzu || zu != zu) v[1] = NAN;
    if (dzl < 0.0) v[1] = nmin(v[1], -tau * zl / dzl);
    if (dzu < 0.0) v[1] = nmin(v[1], -tau * zu / dzu);
  }
  const int op[2] = {0, 0};
  wg_reduce<2>(red, v, op);
  if (threadIdx.x < 2) part[(size_t)threadIdx.x * nwg + wg0 + blockIdx.x] = red[threadIdx.x][0];
}

__global__ __launch_bounds__(256) void k_ip_stats_final(int nwg, const double* __restrict__ part, double* __restrict__ alpha_local) {
  __shared__ double red[2][256];
  double v[2] = {1.0, 1.0};
  for (int i = threadIdx.x; i < nwg; i += 256) { v[0] = nmin(v[0], part[i]); v[1] = nmin(v[1], part[(size_t)nwg + i]); }
  const int op[2] = {0, 0};
  wg_reduce<2>(red, v, op);
  if (threadIdx.x < 2) alpha_local[threadIdx.x] = red[threadIdx.x][0];
}

// ---- the step (interior_point.py:619-626), the barrier diagonals of the next KKT matrix (interface.py:450-465) and the
// elementwise part of the convergence measures (interior_point.py:257-266, 239-244, 286-317) at the NEW iterate.
// Elements: the rows x | s | y_eq | y_link of W (the thread of slack row i also takes y_ineq, s_l, s_u of that row).
__global__ __launch_bounds__(256) void k_ip_step(pp_ip_group g, const double* __restrict__ alpha_table, int nranks, int unified,
                                                 double mu, double* __restrict__ z, const double* __restrict__ dz, int do_z,
                                                 double* __restrict__ part, int wg0, int nwg) {
  __shared__ double red[IP_STEP_SLOTS][256];
  const size_t bpad = (size_t)g.bpad;
  const int nb = ip_nb(g);
  double ap = 0.0, ad = 0.0;
  const bool step = alpha_table != nullptr;
  if (step) {
    ap = alpha_table[0]; ad = alpha_table[1];
    for (int q = 1; q < nranks; ++q) { ap = nmin(ap, alpha_table[2 * q]); ad = nmin(ad, alpha_table[2 * q + 1]); }
    if (unified) ap = ad = nmin(ap, ad);
  }
  // coupling variables: nfs of them, tied by every instance (stochastic programs), or -- mapped groups, time blocks --
  // ncz coupling states, the second half of the coupling solution [d rho | d z]
  const bool mapped = g.zoff != nullptr;
  if (do_z && step && blockIdx.x == 0) {
    const int nz = mapped ? g.ncz : g.nfs;
    const double* dzs = mapped ? dz + g.ncz : dz;
    for (int k = threadIdx.x; k < nz; k += 256) z[k] = z[k] + ap * dzs[k];
  }
  const size_t total = (size_t)(nb - g.mi) * bpad;
  double v[IP_STEP_SLOTS] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    int r = (int)(e / bpad);
    const int b = (int)(e % bpad);
    if (r >= g.n + g.mi + g.me) r += g.mi;         // (the y_ineq rows belong to the slack threads)
    if (b >= g.batch) continue;
    const size_t at = (size_t)r * bpad + b;
    if (r < g.n + g.mi) {
      const VarRows q = var_rows(g, r);
      double x = g.W[at];
      const double lo = g.bounds[q.lo * bpad + b], hi = g.bounds[q.hi * bpad + b];
      double zl = g.W[q.zl * bpad + b], zu = g.W[q.zu * bpad + b];
      if (step) {
        const double dx = g.delta[at];
        const double dzl = (mu - zl * dx) / (x - lo) - zl;
        const double dzu = (mu + zu * dx) / (hi - x) - zu;
        x = x + ap * dx; zl = zl + ad * dzl; zu = zu + ad * dzu;
        g.W[at] = x; g.W[q.zl * bpad + b] = zl; g.W[q.zu * bpad + b] = zu;
      }
      const int srow = r < g.n ? g.src_dp + r : g.src_ds + (r - g.n);
      g.src[(size_t)srow * bpad + b] = zl / (x - lo) + zu / (hi - x);
      if (lo > -INFINITY) { const double c = (x - lo) * zl; v[0] = nmax(v[0], fabs(c)); v[1] = nmax(v[1], fabs(c - mu)); }
      if (hi < INFINITY) { const double c = (hi - x) * zu; v[0] = nmax(v[0], fabs(c)); v[1] = nmax(v[1], fabs(c - mu)); }
      v[3] = v[3] + (fabs(zl) + fabs(zu));
      if (r >= g.n) {
        const size_t ay = (size_t)(g.n + g.mi + g.me + (r - g.n)) * bpad + b;
        double y = g.W[ay];
        if (step) { y = y + ad * g.delta[ay]; g.W[ay] = y; }
        v[2] = nmax(v[2], fabs((-y - zl) + zu));
        v[4] = v[4] + fabs(y);
      }
    } else {
      double y = g.W[at];
      if (step) { y = y + ad * g.delta[at]; g.W[at] = y; }
      v[4] = v[4] + fabs(y);
    }
  }
  if (g.nfw > 0) {
    // the multipliers of the forward link live in the coupling block (sc_ip_interface.py:308-357); the instance keeps a
    // copy behind its bound duals (its gradient rows read it) and moves it with the rho part of the coupling solution
    const size_t yf0 = (size_t)(nb + 2 * g.n + 2 * g.mi);
    const size_t totalf = (size_t)g.nfw * bpad;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < totalf; e += (size_t)gridDim.x * 256) {
      const int k = (int)(e / bpad);
      const int b = (int)(e % bpad);
      if (b >= g.batch) continue;
      const size_t at = (yf0 + k) * bpad + b;
      double y = g.W[at];
      if (step) { y = y + ad * dz[g.zoff[bpad + b] + k]; g.W[at] = y; }
      v[4] = v[4] + fabs(y);
    }
  }
  const int op[IP_STEP_SLOTS] = {1, 1, 1, 2, 2};
  wg_reduce<IP_STEP_SLOTS>(red, v, op);
  if (threadIdx.x < IP_STEP_SLOTS) part[(size_t)threadIdx.x * nwg + wg0 + blockIdx.x] = red[threadIdx.x][0];
}

// ---- the rows that need the scenario data: grad_x L = c + H x + A_eq^T y_eq + A_ineq^T y_ineq + L^T y_link, A_eq x - b,
// A_ineq x - s, x_fs - z.  One row x 64 instances per wave and step, IP_RPW rows per wave; the terms of a row are
// wave-uniform {source row, row of W} pairs, requested four at a time.
// NV instances per lane (2 where the padded batch is a multiple of 128: every operand is one 16-byte load per lane, and
// the scalar record reads, address arithmetic and branches of a row are spent once for 128 instances).
template <int NV>
__global__ __launch_bounds__(256) void k_ip_rows(pp_ip_group g, const double* __restrict__ z, double* __restrict__ part, int wg0,
                                                 int nwg, int rpw, double* __restrict__ cpl) {
  __shared__ double red[IP_ROWS_SLOTS][256];
  const size_t bpad = (size_t)g.bpad;
  const int nchunk = g.bpad / (64 * NV);
  const int chunk = blockIdx.x % nchunk, tile = blockIdx.x / nchunk;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int b = (chunk * 64 + lane) * NV;
  const int nprog = g.n + g.me + g.mi + g.nfs + g.nfw;
  const int nb = ip_nb(g);
  const bool mapped = g.zoff != nullptr;
  const double* __restrict__ W = g.W + b;
  const double* __restrict__ S = g.src + b;
  double v[IP_ROWS_SLOTS] = {0.0, 0.0, 0.0};
  for (int q = 0; q < rpw; ++q) {
    const int slot = (tile * 4 + wave) * rpw + q;         // (wave-uniform)
    if (slot >= nprog) break;
    if (slot == 0 && g.obj_row >= 0) {
      // nonlinear model: the instance's objective value comes from the model's own evaluation (a row of `data`); the
      // gradient rows then hold grad f and the constraint rows -c(x) instead of the linear / constant terms of a QP
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (b + i < g.batch) v[2] = v[2] + g.data[(size_t)g.obj_row * bpad + b + i];
    }
    // rows are worked on in the producer's order (prog[4 slot + 3]): rows that read the same Jacobian entries -- a
    // constraint row and the gradient rows of its variables -- sit next to each other, so the second read of an entry
    // meets the L2 instead of HBM
    const int p = g.prog[4 * slot + 3];
    const int t0 = g.prog[4 * p], tH = g.prog[4 * p + 1], t1 = g.prog[4 * p + 2];
    // what the row needs besides its terms is requested first (it does not depend on them)
    const bool primal = p < g.n;
    int erow = 0;             // row of the subtrahend: c (primal) / b_eq / s; link rows subtract z[k]
    const double* ebase = g.data + b;
    size_t orow;
    if (primal) { erow = p; orow = 0; }
    else if (p < g.n + g.me) { erow = g.n + (p - g.n); orow = (size_t)(g.n + g.mi + (p - g.n)); }
    else if (p < g.n + g.me + g.mi) { erow = g.n + (p - g.n - g.me); ebase = W; orow = (size_t)(g.n + g.mi + g.me + (p - g.n - g.me)); }
    else { erow = -1; orow = (size_t)(g.n + 2 * g.mi + g.me + (p - g.n - g.me - g.mi)); }
    // link rows: the nfs rows inside the block, then (time blocks) the nfw rows of the forward link, whose residual
    // belongs to the coupling block of the right-hand side
    const int kk = p - g.n - g.me - g.mi;
    const bool forward = erow < 0 && kk >= g.nfs;
    double ev[NV], zl[NV], zu[NV], xp[NV];
    int zo[NV];
    if (erow >= 0) ldv<NV>(ebase + (size_t)erow * bpad, ev);
    else if (!mapped) { const double zk = z[kk];
#pragma unroll
      for (int i = 0; i < NV; ++i) ev[i] = zk; }
    else {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        zo[i] = g.zoff[(forward ? bpad : (size_t)0) + b + i] + (forward ? kk - g.nfs : kk);
        ev[i] = z[zo[i]];
      }
    }
    if (primal) {
      ldv<NV>(W + (size_t)(nb + p) * bpad, zl);
      ldv<NV>(W + (size_t)(nb + g.n + p) * bpad, zu);
      ldv<NV>(W + (size_t)p * bpad, xp);
    }
    double accH[NV], acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { accH[i] = 0.0; acc[i] = 0.0; }
    int t = t0;
    for (; t + 4 <= t1; t += 4) {
      int s[4], w[4];
      double a[4][NV], c[4][NV];
#pragma unroll
      for (int k = 0; k < 4; ++k) { s[k] = g.terms[2 * (t + k)]; w[k] = g.terms[2 * (t + k) + 1]; }
#pragma unroll
      for (int k = 0; k < 4; ++k) { ldv<NV>(W + (size_t)w[k] * bpad, c[k]); ldv<NV>(S + (size_t)(s[k] < 0 ? 0 : s[k]) * bpad, a[k]); }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const double term = s[k] < 0 ? c[k][i] : a[k][i] * c[k][i];
          if (t + k < tH) accH[i] = accH[i] + term; else acc[i] = acc[i] + term;
        }
      }
    }
    for (; t < t1; ++t) {
      const int s1 = g.terms[2 * t], w1 = g.terms[2 * t + 1];
      double c[NV], a[NV];
      ldv<NV>(W + (size_t)w1 * bpad, c);
      ldv<NV>(S + (size_t)(s1 < 0 ? 0 : s1) * bpad, a);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const double term = s1 < 0 ? c[i] : a[i] * c[i];
        if (t < tH) accH[i] = accH[i] + term; else acc[i] = acc[i] + term;
      }
    }
    if (primal) {
      double G[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const double gH = ev[i] + accH[i];                  // gradient of the objective (interior_point.py:192)
        G[i] = gH + acc[i];
        if (b + i < g.batch) {
          v[1] = nmax(v[1], fabs((G[i] - zl[i]) + zu[i]));
          if (g.obj_row < 0) v[2] = v[2] + xp[i] * (0.5 * accH[i] + ev[i]);      // 1/2 x'Hx + c'x
        }
      }
      stv<NV>(g.G + (size_t)p * bpad + b, G);
    } else {
      double out[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const double res = acc[i] - ev[i];
        out[i] = -res;
        if (b + i < g.batch) v[0] = nmax(v[0], fabs(res));
      }
      if (forward) {
#pragma unroll
        for (int i = 0; i < NV; ++i) if (b + i < g.batch) cpl[zo[i]] = out[i];
      } else stv<NV>(g.rhs + orow * bpad + b, out);
    }
  }
  const int op[IP_ROWS_SLOTS] = {1, 1, 2};
  wg_reduce<IP_ROWS_SLOTS>(red, v, op);
  if (threadIdx.x < IP_ROWS_SLOTS) part[(size_t)threadIdx.x * nwg + wg0 + blockIdx.x] = red[threadIdx.x][0];
}

// ---- mapped groups: the link duals of an instance go to the coupling states of ITS two links (the z rows of the coupling
// right-hand side, sc_ip_interface.py:857-861).  A coupling state has one backward and one forward link: at most two addends
// on a zeroed entry, so the order of the atomic adds does not show in the sum.
__global__ __launch_bounds__(256) void k_ip_links_mapped(pp_ip_group g, double* __restrict__ zpart) {
  const size_t bpad = (size_t)g.bpad;
  const size_t total = (size_t)(g.nfs + g.nfw) * bpad;
  const size_t yb0 = (size_t)(g.n + 2 * g.mi + g.me), yf0 = (size_t)(ip_nb(g) + 2 * g.n + 2 * g.mi);
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int k = (int)(e / bpad);
    const int b = (int)(e % bpad);
    if (b >= g.batch) continue;
    const bool fw = k >= g.nfs;
    const double y = g.W[(fw ? yf0 + (k - g.nfs) : yb0 + k) * bpad + b];
    unsafeAtomicAdd(zpart + g.zoff[(fw ? bpad : (size_t)0) + b] + (fw ? k - g.nfs : k), y);
  }
}

// ---- this rank's scalars: workgroups 0 .. 7 combine one slot each of the partials of k_ip_step and k_ip_rows (fixed order:
// deterministic); the others sum the link duals of every coupling variable over the instances of all groups (one row per
// wave; the coupling block of the right-hand side, sc_ip_interface.py:1694-1696, and of grad L)
struct IpLinks { int ng, nfs; const double* ylink[IP_MAXG]; int batch[IP_MAXG], bpad[IP_MAXG]; };
__global__ __launch_bounds__(256) void k_ip_local(IpLinks L, int nwg_step, const double* __restrict__ part_step, int nwg_rows,
                                                  const double* __restrict__ part_rows, double* __restrict__ v_local) {
  __shared__ double red[1][256];
  if (blockIdx.x < 8) {
    // slot -> entry of v_local: {primal inf, |grad_x L|, compl(0), compl(mu), sum |z|, sum |y|, objective, |grad_s L|}
    const int slot = blockIdx.x;
    const bool from_rows = slot == 0 || slot == 1 || slot == 6;
    const int src_slot = slot == 0 ? 0 : slot == 1 ? 1 : slot == 6 ? 2 : slot == 2 ? 0 : slot == 3 ? 1 : slot == 7 ? 2 : slot == 4 ? 3 : 4;
    const int n = from_rows ? nwg_rows : nwg_step;
    const double* p = (from_rows ? part_rows : part_step) + (size_t)src_slot * n;
    const bool sum = slot >= 4 && slot <= 6;
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) v = sum ? v + p[i] : nmax(v, p[i]);
    const int op[1] = {sum ? 2 : 1};
    wg_reduce<1>(red, &v, op);
    if (threadIdx.x == 0) v_local[slot] = red[0][0];
    return;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int k = ((int)blockIdx.x - 8) * 4 + wave;
  if (k >= L.nfs) return;
  double s = 0.0;
  for (int gi = 0; gi < L.ng; ++gi) {
    const double* y = L.ylink[gi] + (size_t)k * L.bpad[gi];
    for (int b = lane; b < L.batch[gi]; b += 64) s = s + y[b];
  }
  for (int off = 32; off > 0; off >>= 1) s = s + __shfl_down(s, off, 64);
  if (lane == 0) v_local[8 + k] = s;
}

// ---- all ranks' scalars -> coupling right-hand side + mailbox.  Rows are combined in rank order on every rank: every rank
// publishes the same numbers (the loop's control flow must agree).  The ncoup entries behind the 8 scalars of a rank are
// its part of the coupling right-hand side; from entry dual_from on they are also -grad L of the coupling variables.
__device__ __forceinline__ double couple_entry(const double* __restrict__ v_table, int nranks, int nv, int k) {
  double s = 0.0;
  for (int r = 0; r < nranks; ++r) s = s + v_table[(size_t)r * nv + 8 + k];
  return s;
}

// long coupling blocks (time-staged problems): the sums on many workgroups, one maximum per workgroup for k_ip_publish
__global__ __launch_bounds__(256) void k_ip_couple(const double* __restrict__ v_table, int nranks, int ncoup, int dual_from,
                                                   double* __restrict__ rhs_coupling, double* __restrict__ wgmax) {
  __shared__ double red[1][256];
  const int nv = 8 + ncoup;
  double m = 0.0;
  for (int k = blockIdx.x * 256 + threadIdx.x; k < ncoup; k += gridDim.x * 256) {
    const double s = couple_entry(v_table, nranks, nv, k);
    rhs_coupling[k] = s;
    if (k >= dual_from) m = nmax(m, fabs(s));
  }
  const int op[1] = {1};
  wg_reduce<1>(red, &m, op);
  if (threadIdx.x == 0) wgmax[blockIdx.x] = red[0][0];
}

__global__ __launch_bounds__(256) void k_ip_publish(const double* __restrict__ v_table, const double* __restrict__ alpha_table,
                                                    int nranks, int ncoup, int dual_from, double* __restrict__ rhs_coupling,
                                                    const double* __restrict__ wgmax, int nwgmax, double* mail, long long seq) {
  __shared__ double red[1][256];
  const int nv = 8 + ncoup;
  double m = 0.0;
  if (wgmax) {
    for (int k = threadIdx.x; k < nwgmax; k += 256) m = nmax(m, wgmax[k]);
  } else {
    for (int k = threadIdx.x; k < ncoup; k += 256) {
      const double s = couple_entry(v_table, nranks, nv, k);
      rhs_coupling[k] = s;
      if (k >= dual_from) m = nmax(m, fabs(s));           // |grad_z L| = |-sum of the link duals|
    }
  }
  const int op[1] = {1};
  wg_reduce<1>(red, &m, op);
  if (threadIdx.x == 0) {
    double o[9] = {0.0, red[0][0], 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 1.0};
    for (int r = 0; r < nranks; ++r) {
      const double* v = v_table + (size_t)r * nv;
      o[0] = nmax(o[0], v[0]); o[1] = nmax(o[1], nmax(v[1], v[7])); o[2] = nmax(o[2], v[2]); o[3] = nmax(o[3], v[3]);
      o[4] = o[4] + v[4]; o[5] = o[5] + v[5]; o[6] = o[6] + v[6];
      if (alpha_table) { o[7] = nmin(o[7], alpha_table[2 * r]); o[8] = nmin(o[8], alpha_table[2 * r + 1]); }
    }
    for (int i = 0; i < 9; ++i) mail[i] = o[i];
    __threadfence_system();
    __hip_atomic_store((long long*)(mail + 15), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

int ip_check(pp_handle h, int ngroups, const pp_ip_group* g, const char* what) {
  if (!h || ngroups < 1 || ngroups > IP_MAXG || !g) return fail(h, 3, std::string(what) + ": bad arguments (1 to 8 groups)");
  for (int i = 0; i < ngroups; ++i) {
    const pp_ip_group& q = g[i];
    if (q.n < 0 || q.mi < 0 || q.me < 0 || q.nfs < 0 || q.batch < 1 || q.bpad < q.batch || (q.bpad & 63) ||
        !q.W || !q.bounds || !q.data || !q.src || !q.G || !q.rhs || !q.prog || !q.terms)
      return fail(h, 3, std::string(what) + ": inconsistent group descriptor");
    // either every instance ties all nfs coupling variables (no map, the same nfs in every group), or the groups are
    // mapped: per-instance offsets into ncz coupling states, forward links allowed
    const bool mapped = q.zoff != nullptr;
    if (mapped != (g[0].zoff != nullptr) || (!mapped && (q.nfs != g[0].nfs || q.nfw != 0)) ||
        (mapped && (q.ncz != g[0].ncz || q.ncz < 0 || q.nfw < 0 || q.nfs > q.ncz || q.nfw > q.ncz)))
      return fail(h, 3, std::string(what) + ": inconsistent coupling description of the groups");
  }
  return 0;
}

// device scratch for the partials of one call (grows, never shrinks; stream-ordered reuse)
int ip_scratch(pp_handle h, size_t doubles) {
  if (h->ip_part_cap >= doubles) return 0;
  PP_HIP(hipStreamSynchronize(h->stream));
  void* p = nullptr;
  if (hipMalloc(&p, doubles * sizeof(double)) != hipSuccess) return fail(h, 1, "hipMalloc failed (interior-point scratch)");
  // (the partials k_ip_step left for pp_ip_residuals survive a growth of the buffer)
  if (h->ip_part && h->ip_part_cap) PP_HIP(hipMemcpy(p, h->ip_part, h->ip_part_cap * sizeof(double), hipMemcpyDeviceToDevice));
  if (h->ip_part) (void)hipFree(h->ip_part);
  h->ip_part = (double*)p;
  h->ip_part_cap = doubles;
  return 0;
}

unsigned ew_grid(const pp_ip_group& g, int rows) {
  return (unsigned)std::min<size_t>(IP_EW_MAXWG, ((size_t)rows * g.bpad + 255) / 256);
}
int rows_nv(const pp_ip_group& g) { return (g.bpad % 128 == 0) ? 2 : 1; }      // instances per lane of k_ip_rows
int rows_rpw(const pp_ip_group& g) {       // rows per wave of k_ip_rows
  const size_t nprog = (size_t)(g.n + g.me + g.mi + g.nfs + g.nfw), nchunk = (size_t)(g.bpad / (64 * rows_nv(g)));
  return (int)std::max<size_t>(IP_RPW, (nprog * nchunk + 4 * IP_ROWS_MAXWG - 1) / (4 * IP_ROWS_MAXWG));
}
unsigned rows_grid(const pp_ip_group& g) {
  const int nprog = g.n + g.me + g.mi + g.nfs + g.nfw, rpw = rows_rpw(g);
  return (unsigned)((nprog + 4 * rpw - 1) / (4 * rpw)) * (unsigned)(g.bpad / (64 * rows_nv(g)));
}

// Workgroups of the three reducing kernels over these groups, and the scratch they share: the partials of k_ip_step
// [IP_STEP_SLOTS][nwg_step], of k_ip_rows [IP_ROWS_SLOTS][nwg_rows] and of k_ip_stats [2][nwg_stats], in this order.
struct IpSizes { unsigned nwg_step = 0, nwg_rows = 0, nwg_stats = 0; };
int ip_sizes(pp_handle h, int ngroups, const pp_ip_group* g, IpSizes* out) {
  IpSizes z;
  for (int i = 0; i < ngroups; ++i) {
    z.nwg_step += ew_grid(g[i], g[i].n + g[i].mi + g[i].me + g[i].nfs);
    z.nwg_rows += rows_grid(g[i]);
    z.nwg_stats += ew_grid(g[i], g[i].n + g[i].mi);
  }
  *out = z;
  return ip_scratch(h, (size_t)IP_STEP_SLOTS * z.nwg_step + (size_t)IP_ROWS_SLOTS * z.nwg_rows + (size_t)2 * z.nwg_stats);
}

}  // namespace

extern "C" {

int pp_ip_rhs(pp_handle h, int ngroups, const pp_ip_group* g, double mu) {
  if (int rc = ip_check(h, ngroups, g, "pp_ip_rhs")) return rc;
  PP_HIP(hipSetDevice(h->device));
  PhaseScope ps(h, PP_NPHASE_SOLVER + 0, ngroups);
  for (int i = 0; i < ngroups; ++i)
    if (g[i].n + g[i].mi > 0) hipLaunchKernelGGL(k_ip_rhs, dim3(ew_grid(g[i], g[i].n + g[i].mi)), dim3(256), 0, h->stream, g[i], mu);
  PP_HIP(hipGetLastError());
  return 0;
}

int pp_ip_step_lengths(pp_handle h, int ngroups, const pp_ip_group* g, double tau, double mu, double* alpha_local) {
  if (int rc = ip_check(h, ngroups, g, "pp_ip_step_lengths")) return rc;
  if (!alpha_local) return fail(h, 3, "pp_ip_step_lengths: no output array");
  PP_HIP(hipSetDevice(h->device));
  for (int i = 0; i < ngroups; ++i)
    if (!g[i].delta) return fail(h, 3, "pp_ip_step_lengths: no step (delta)");
  IpSizes sz;
  if (int rc = ip_sizes(h, ngroups, g, &sz)) return rc;
  const unsigned nwg = sz.nwg_stats;
  PhaseScope ps(h, PP_NPHASE_SOLVER + 1, ngroups + 1);
  double* part = h->ip_part + (size_t)IP_STEP_SLOTS * sz.nwg_step + (size_t)IP_ROWS_SLOTS * sz.nwg_rows;
  unsigned wg0 = 0;
  for (int i = 0; i < ngroups; ++i) {
    const unsigned n = ew_grid(g[i], g[i].n + g[i].mi);
    if (n) hipLaunchKernelGGL(k_ip_stats, dim3(n), dim3(256), 0, h->stream, g[i], tau, mu, part, (int)wg0, (int)nwg);
    wg0 += n;
  }
  hipLaunchKernelGGL(k_ip_stats_final, dim3(1), dim3(256), 0, h->stream, (int)nwg, part, alpha_local);
  PP_HIP(hipGetLastError());
  return 0;
}

int pp_ip_take_step(pp_handle h, int ngroups, const pp_ip_group* g, const double* alpha_table, int nranks, int unified, double mu,
                    double* z, const double* dz) {
  if (int rc = ip_check(h, ngroups, g, "pp_ip_take_step")) return rc;
  const int ncoupled = g[0].zoff ? g[0].ncz : g[0].nfs;
  if (alpha_table && (nranks < 1 || (ncoupled > 0 && (!z || !dz)))) return fail(h, 3, "pp_ip_take_step: bad arguments");
  PP_HIP(hipSetDevice(h->device));
  for (int i = 0; i < ngroups; ++i)
    if (alpha_table && !g[i].delta) return fail(h, 3, "pp_ip_take_step: no step (delta)");
  IpSizes sz;
  if (int rc = ip_sizes(h, ngroups, g, &sz)) return rc;
  const unsigned nwg_s = sz.nwg_step;
  h->ip_step_done = true;
  // pp_ip_residuals combines the partials left here: it must see the same slot layout (same groups, same grids)
  h->ip_step_layout[0] = sz.nwg_step; h->ip_step_layout[1] = sz.nwg_rows; h->ip_step_layout[2] = sz.nwg_stats;
  PhaseScope ps(h, PP_NPHASE_SOLVER + 2, ngroups);
  unsigned wg0 = 0;
  bool z_moved = false;                      // the coupling variables move with the first launch that has a grid
  for (int i = 0; i < ngroups; ++i) {
    const unsigned n = ew_grid(g[i], g[i].n + g[i].mi + g[i].me + g[i].nfs);
    if (n) hipLaunchKernelGGL(k_ip_step, dim3(n), dim3(256), 0, h->stream, g[i], alpha_table, nranks, unified, mu, z, dz,
                              z_moved ? 0 : 1, h->ip_part, (int)wg0, (int)nwg_s);
    if (n) z_moved = true;
    wg0 += n;
  }
  PP_HIP(hipGetLastError());
  return 0;
}

int pp_ip_residuals(pp_handle h, int ngroups, const pp_ip_group* g, const double* z, double* v_local) {
  if (int rc = ip_check(h, ngroups, g, "pp_ip_residuals")) return rc;
  const bool mapped = g[0].zoff != nullptr;
  if (!v_local || ((mapped ? g[0].ncz : g[0].nfs) > 0 && !z)) return fail(h, 3, "pp_ip_residuals: bad arguments");
  PP_HIP(hipSetDevice(h->device));
  if (!h->ip_step_done) return fail(h, 3, "pp_ip_residuals: call pp_ip_take_step first");
  IpSizes sz;
  if (int rc = ip_sizes(h, ngroups, g, &sz)) return rc;
  const unsigned nwg_s = sz.nwg_step, nwg_r = sz.nwg_rows;
  if (h->ip_step_layout[0] != sz.nwg_step || h->ip_step_layout[1] != sz.nwg_rows || h->ip_step_layout[2] != sz.nwg_stats)
    return fail(h, 3, "pp_ip_residuals: not the group list pp_ip_take_step was called with (the step partials have another layout)");
  PhaseScope ps(h, PP_NPHASE_SOLVER + 3, ngroups + 1);
  double* part_rows = h->ip_part + (size_t)IP_STEP_SLOTS * nwg_s;
  unsigned wg0 = 0;
  IpLinks L;
  L.ng = ngroups; L.nfs = mapped ? 0 : g[0].nfs;
  // mapped groups: this rank's part of the coupling right-hand side [rho rows | z rows] starts from zero; the row kernels
  // store the residuals of the forward links, k_ip_links_mapped adds the link duals
  double* cpl = v_local + 8;
  if (mapped && g[0].ncz > 0) PP_HIP(hipMemsetAsync(cpl, 0, (size_t)2 * g[0].ncz * sizeof(double), h->stream));
  for (int i = 0; i < ngroups; ++i) {
    const unsigned n = rows_grid(g[i]);
    if (n && rows_nv(g[i]) == 2) hipLaunchKernelGGL(k_ip_rows<2>, dim3(n), dim3(256), 0, h->stream, g[i], z, part_rows, (int)wg0, (int)nwg_r, rows_rpw(g[i]), cpl);
    else if (n) hipLaunchKernelGGL(k_ip_rows<1>, dim3(n), dim3(256), 0, h->stream, g[i], z, part_rows, (int)wg0, (int)nwg_r, rows_rpw(g[i]), cpl);
    if (mapped && g[i].nfs + g[i].nfw > 0)
      hipLaunchKernelGGL(k_ip_links_mapped, dim3(ew_grid(g[i], g[i].nfs + g[i].nfw)), dim3(256), 0, h->stream, g[i], cpl + g[0].ncz);
    wg0 += n;
    L.ylink[i] = g[i].W + (size_t)(g[i].n + 2 * g[i].mi + g[i].me) * g[i].bpad;
    L.batch[i] = g[i].batch; L.bpad[i] = g[i].bpad;
  }
  hipLaunchKernelGGL(k_ip_local, dim3(8 + (L.nfs + 3) / 4), dim3(256), 0, h->stream, L, (int)nwg_s, h->ip_part, (int)nwg_r,
                     part_rows, v_local);
  PP_HIP(hipGetLastError());
  return 0;
}

int pp_ip_publish(pp_handle h, const double* v_table, const double* alpha_table, int nranks, int ncoup, int dual_from,
                  double* rhs_coupling) {
  if (!h || !v_table || nranks < 1 || ncoup < 0 || dual_from < 0 || dual_from > ncoup || (ncoup > 0 && !rhs_coupling))
    return fail(h, 3, "pp_ip_publish: bad arguments");
  PP_HIP(hipSetDevice(h->device));
  if (!h->ip_mail_host) {
    void* hp = nullptr;
    void* dp = nullptr;
    PP_HIP(hipHostMalloc(&hp, 16 * sizeof(double), hipHostMallocMapped));
    std::memset(hp, 0, 16 * sizeof(double));
    PP_HIP(hipHostGetDevicePointer(&dp, hp, 0));
    h->ip_mail_host = (volatile double*)hp;
    h->ip_mail_dev = (double*)dp;
    h->ip_seq = 0;
  }
  const double* wgmax = nullptr;
  int nwgmax = 0;
  if (ncoup > 2048) {
    constexpr int MAXWG = 256;
    if (!h->ip_cmax) {
      void* p = nullptr;
      if (hipMalloc(&p, MAXWG * sizeof(double)) != hipSuccess) return fail(h, 1, "hipMalloc failed (interior-point scratch)");
      h->ip_cmax = (double*)p;
    }
    nwgmax = std::min(MAXWG, (ncoup + 255) / 256);
    hipLaunchKernelGGL(k_ip_couple, dim3(nwgmax), dim3(256), 0, h->stream, v_table, nranks, ncoup, dual_from, rhs_coupling,
                       h->ip_cmax);
    wgmax = h->ip_cmax;
  }
  hipLaunchKernelGGL(k_ip_publish, dim3(1), dim3(256), 0, h->stream, v_table, alpha_table, nranks, ncoup, dual_from, rhs_coupling,
                     wgmax, nwgmax, h->ip_mail_dev, ++h->ip_seq);
  PP_HIP(hipGetLastError());
  return 0;
}

int pp_ip_wait(pp_handle h, double out[10]) {
  if (!h || !out || !h->ip_mail_host) return fail(h, 3, "pp_ip_wait before pp_ip_publish");
  PP_HIP(hipSetDevice(h->device));
  const long long want = h->ip_seq;
  const long long* seqp = (const long long*)(h->ip_mail_host + 15);
  bool seen = false;
  const auto t0 = std::chrono::steady_clock::now();
  for (long spin = 0;; ++spin) {
    if (__atomic_load_n(seqp, __ATOMIC_ACQUIRE) == want) { seen = true; break; }
    if ((spin & 1023) == 1023 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.25) break;
  }
  if (!seen) {
    PP_HIP(hipStreamSynchronize(h->stream));       // (also surfaces an asynchronous device error instead of spinning on it)
    if (__atomic_load_n(seqp, __ATOMIC_ACQUIRE) != want) return fail(h, 3, "pp_ip_wait: the mailbox was not written");
  }
  for (int i = 0; i < 9; ++i) out[i] = h->ip_mail_host[i];
  out[9] = 0.0;
  return 0;
}

}  // extern "C"
