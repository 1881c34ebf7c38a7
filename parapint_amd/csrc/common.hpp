// Shared declarations of the translation units of libparapint_hip.so (MI355X / gfx950):
//
//   factor.hip   leaf assembly + left-looking block LDL^T in L form: gather / scale kernels, root front
//   schur.hip    Schur update on the fp64 matrix cores, inertia counts, dispatch of the factorisation of S
//   dense.hip    dense LDL^T / Bunch-Kaufman of S, dense coupling solve
//   bcr.hip      block-tridiagonal S: block cyclic reduction and its solve
//   solve.hip    forward / backward substitution of the blocks, coupling rows
//   vecops.hip   vector kernels of the interior-point step
//   api.hip      handle life cycle, symbolic phase (plan -> device images), uploads, bindings, statistics, host staging
//
// Mapping (see plan.hpp): lane = scenario block.  Every value array is [entry][instance], so a wavefront touches 64
// consecutive doubles (512 B) per access and all control flow / index data is wave-uniform (scalar loads, scalar
// branches): the sparse phase is a pure HBM / L2 streaming workload with no divergence.  One 64-thread workgroup = one
// task x 64 instances.
#pragma once
#include <hip/hip_runtime.h>

#include <array>
#include <atomic>
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


namespace ppd {


constexpr int WAVE = 64;
typedef double double4_t __attribute__((ext_vector_type(4)));   // accumulator of v_mfma_f64_16x16x4
constexpr int PP_MAX_SPLIT = 8;
constexpr int PP_MT_WIDE_NC = 512;   // from this coupling dimension on the Schur update works on 32 x 32 super-tiles (k_schur_mfma_wide)
constexpr int PP_SCHUR_SLICES = 4;   // mapped groups: workgroups that share the records of one Schur tile (partial cliques in slots of Sloc)
constexpr int PP_MT_SLICE = 4;  // records (panel columns) of one 16 x 16 Schur tile per work item of k_schur_mfma
constexpr int PP_CSLOTS = 64;  // the inertia / growth counters are kept in this many slots of 4 ints, summed by the tail writer
constexpr int PP_TAIL = 8;    // doubles behind the n_c x n_c Schur block: zero pivots, pos, neg, host failures, growth, reserved
constexpr int PP_NPHASE = 12;       // assemble, factor, schur, dense, fwd, fwd_coupling, coupling_solve, bwd (pp_phase_times) +
constexpr int PP_NPHASE_SOLVER = 8; // the interior-point step: right-hand side, step lengths, step, residuals + scalars (pp_ip_phase_times)
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
  const int *ttask, *trec;            // tile tasks (plan.hpp, kind 4) and their source-panel records
  const int *chain_col;               // chain fronts: the columns of every front (new index; the native backward sweep gets the caller's rows)
  const int *chain_hdr, *chain_pan;   // chain fronts (plan.hpp): per front {m, W, panels, first panel record}, per panel {piv, w, uoff, boff, doff, sub, col0, f}
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
// ------------------------------------------------------------------------------------------
// inputs of a plan that pp_end_symbolic still has to build (pp_add_group* copies them)
struct PendingPlan {
  int n = 0;
  bool have_vals = false;
  pp::PlanOptions opt;
  std::vector<int32_t> rowK, colK, rowB, colB;
  std::vector<double> rep_vals;
};

struct Group {
  PendingPlan* pending = nullptr;
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
  const int* chain_colN = nullptr;  // chain fronts: caller's row of every front column (native vectors)
  std::vector<size_t> chain_lds;   // per factor level: dynamic LDS bytes of its chain fronts (k_chain_front), 0 if none
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
  // host boundary: rows of the compact device input that mirror the caller's pinned staging array (same values on both
  // sides): a later staging pass sends only the column ranges whose values changed (api.hip: StageJob)
  std::vector<uint8_t> staged_row_valid;
  const double* stage_host = nullptr;
  std::vector<int> cmap_host;        // mapped group: [batch][nc_loc] global coupling indices
  // a-posteriori check of a back-solve (refine.hip): canonical pattern (kept from the symbolic phase), residual records by row of
  // [K | A^T] over the transposed input (vraw) / over the producer's sources (vsrc, csrc), border records by coupling row,
  // per-instance maxima, residual and correction vectors of a refinement step (value storage, allocated at first use)
  std::vector<int32_t> pat_rowK, pat_colK, pat_rowB, pat_colB;
  const int *res_ptr = nullptr, *res_vraw = nullptr, *res_xnew = nullptr, *res_xold = nullptr;
  const int *res_brow_new = nullptr, *res_brow_old = nullptr;      // row (elimination order / caller's order) of every position of the execution order
  const int *res_bptr = nullptr, *res_bvraw = nullptr, *res_bxnew = nullptr, *res_bxold = nullptr;
  int res_ne = 0, res_bne = 0, res_nrows = 0;
  std::vector<int> res_vraw_host, res_bvraw_host;
  int *res_vsrc = nullptr, *res_bvsrc = nullptr;
  double *res_csrc = nullptr, *res_bcsrc = nullptr;
  unsigned long long *res_rmax = nullptr, *res_smax = nullptr;
  double *res_bpart = nullptr, *res_R = nullptr, *res_D = nullptr;
  bool last_fused = false;           // the last factorisation read the sources through the entry records (else the transposed input)
  const double* save_rhs_native = nullptr;
  double* save_x_native = nullptr;
};


}  // namespace ppd

using namespace ppd;

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

// Host threads of the host boundary (staging of the matrix values, right-hand-side rows): started once per handle,
// parked on a condition variable in between (eight slices x sixteen thread starts per call cost more than the copies they
// made).  start() returns at once -- the caller issues the copy of every finished slice -- and wait() joins the job.
struct StagePool {
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv, cv_done;
  std::function<void()> job;
  long long gen = 0;
  int nwork = 0, pending = 0;
  bool stop = false;
  void worker(int k) {
    long long seen = 0;
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return stop || (gen != seen && k < nwork); });
        if (stop) return;
        seen = gen;
        f = job;
      }
      f();
      {
        std::lock_guard<std::mutex> lk(m);
        if (--pending == 0) cv_done.notify_all();
      }
    }
  }
  // f runs on up to n threads (it claims its work items itself); returns the number of threads that run it (0: none
  // could be started -- the caller runs f itself)
  int start(int n, const std::function<void()>& f) {
    try {
      while ((int)th.size() < n) { const int k = (int)th.size(); th.emplace_back([this, k] { worker(k); }); }
    } catch (...) {
    }
    n = std::min(n, (int)th.size());
    if (n == 0) return 0;
    {
      std::lock_guard<std::mutex> lk(m);
      job = f; nwork = n; pending = n; ++gen;
    }
    cv.notify_all();
    return n;
  }
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [&] { return pending == 0; });
  }
  ~StagePool() {
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
  // dense phase beside the forward sweep (dense.hip): the one-workgroup factorisation of S runs on a stream of its own,
  // forked behind the Schur reduction; whoever reads the factor or writes S next joins it (join_dense)
  hipStream_t dense_stream = nullptr;
  hipEvent_t ev_dense_fork = nullptr, ev_dense_done = nullptr;
  bool dense_pending = false;
  bool schur_on_side = false;    // the Schur update of this factorisation runs on dense_stream (pp_numeric_schur_ex): the dense phase follows it there
  bool dense_overlap = pp::env_switch("PP_NO_DENSE_OVERLAP") == nullptr;   // (measurement switch)
  // interior-point step on device-resident iterates (ipstep.hip): partials of its reductions, pinned mailbox
  double* ip_part = nullptr;
  size_t ip_part_cap = 0;
  double* ip_cmax = nullptr;   // per-workgroup maxima of k_ip_couple (long coupling blocks)
  hipEvent_t ev_coll_side = nullptr;        // behind a collective enqueued on the side stream (pp_allreduce_schur with the Schur update there)
  bool coll_side_pending = false;
  bool ip_step_done = false;
  unsigned ip_step_layout[3] = {0, 0, 0};   // workgroup counts (step, rows, stats) of the last pp_ip_take_step
  volatile double* ip_mail_host = nullptr;
  double* ip_mail_dev = nullptr;
  long long ip_seq = 0;
  // coupling structure: dense S (default) or block-tridiagonal with G blocks of gs rows (n_c = G * gs)
  int btd = 0, gs = 0, G = 0;
  double *btd_fac = nullptr, *btd_inv = nullptr, *btd_x = nullptr, *btd_q = nullptr, *btd_vec = nullptr;
  double *btd_klo = nullptr, *btd_kup = nullptr, *btd_ylo = nullptr, *btd_yup = nullptr;
  int *btd_ipiv = nullptr, *btd_info = nullptr, *scatter_err = nullptr, *btd_elim = nullptr;
  std::vector<int> bcr_off, bcr_ne, bcr_s, bcr_lo;   // per level: offset into btd_elim, eliminated blocks, stride, lower neighbour live
  int btd_sequential = 0;
  bool bcr_lds_attr = false, bcr_ldl_attr = false, dn_lds_attr = false, chain_lds_attr = false, chain_bwd_attr = false;
  double* dn_z = nullptr;        // fat-panel dense factor (n_c > 512): inverted diagonal blocks + work vectors of the panel solve (dense.hip)
  // largest multiplier the unpivoted block factorisation of the cyclic reduction accepts (1 / u, u = 0.01; PP_BCR_LBOUND:
  // test switch -- a bound below 1 sends some blocks to Bunch-Kaufman and leaves others on the unpivoted path)
  double bcr_lbound = pp::env_switch("PP_BCR_LBOUND") ? std::atof(pp::env_switch("PP_BCR_LBOUND")) : 100.0;
  double growth_bound = 1e8;     // 1 / u_rt: a factor entry beyond it flags its instance
  bool growth_fatal = false;     // flagged instances make the factorisation report status 2 (else they are only counted)
  double pivot_threshold = 0.0;  // symbolic-time threshold u for groups added afterwards (0: plan default)
  bool no_fused_sources = pp::env_switch("PP_NO_FUSED_SOURCES") != nullptr;   // measurement switch: assemble the sources first
  bool schur_mfma = pp::env_switch("PP_NO_SCHUR_MFMA") == nullptr;   // MFMA form of the Schur update of unmapped groups (measurement switch)
  bool enqueue_threads = pp::env_switch("PP_NO_ENQUEUE_THREADS") == nullptr;   // one enqueuing host thread per group stream (measurement switch)
  EnqueuePool pool;
  StagePool stage_pool;
  void* stage_job = nullptr;             // staging job in flight (pp_stage_upload_verified_begin .. pp_stage_upload_end), api.hip
  std::vector<hipEvent_t> dl_events;      // one per slice of a staged download (pp_download_solution_rows)
  long long* corner_pos = nullptr;      // sparse Q of a block-tridiagonal S: positions in the Schur layout, values
  double* corner_val = nullptr;
  size_t corner_cap = 0;
  long long corner_nnz = -1;            // pairs of the last pp_factor_schur_corner (the a-posteriori check reads them), -1: none
  hipEvent_t ev_corner_up = nullptr, ev_corner_done = nullptr;
  hipStream_t up_stream = nullptr;
  bool corner_used = false;
  std::mutex alloc_mu, err_mu;
  bool group_streams = pp::env_switch("PP_NO_GROUP_STREAMS") == nullptr;   // pattern groups side by side on streams of their own (measurement switch)
  // forward sweep of an announced right-hand side behind the block factorisation of its own group instead of behind the
  // Schur update and the factorisation of S (pp_solve_forward_ex; measurement switch)
  bool fwd_early = pp::env_switch("PP_NO_EARLY_FORWARD") == nullptr;
  hipEvent_t ev_blocks_done = nullptr;   // recorded on the handle's stream behind the join of the groups' factorisations
  bool blocks_done_valid = false;
  bool dense_dpp = pp::env_switch("PP_NO_DENSE_DPP") == nullptr;           // row broadcasts by DP-ALU DPP in k_ldl_regs (measurement switch)
  bool lane_pairs = pp::env_switch("PP_NO_LANE_PAIRS") == nullptr;   // two instances per lane in the gather kernels (measurement switch)
  double shift_w = 0.0, shift_c = 0.0;   // diagonal shifts of the current pp_numeric_local_shifted call (else 0)
  double mem_factor = 1.0;
  int64_t mem_budget = 0;        // bytes of device value storage the handle may allocate (0: no limit); scaled by mem_factor
  int64_t mem_required = 0;      // bytes of value storage the current plan needs
  bool values_allocated = false;
  std::string err;
  // a-posteriori check (refine.hip): pinned mailbox {rho, group, slot, sequence | x_c, sum A x, sum |A||x|, b_c}
  volatile double* resid_host = nullptr;
  double *resid_dev = nullptr, *resid_best = nullptr, *resid_ax = nullptr, *resid_rc = nullptr, *xc_save = nullptr;
  long long resid_seq = 0;
  bool refining = false, resid_rc_valid = false, resid_on_device = false;
  const double* last_rc = nullptr;   // coupling right-hand side (device) of the last pp_solve_coupling(_dev), or null
  bool have_Q = false;           // the last dense factorisation of S had a coupling block Q (resident in Qd)
  void* rccl_comm = nullptr;     // ncclComm_t of pp_comm_init (api.hip), or null
  int rccl_ranks = 0, rccl_rank = 0;
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


// Entry points between the translation units (not part of the C ABI)
int ppi_dense_factor_schur(pp_handle h, const double* Q_host);                       // dense.hip
int ppi_dense_coupling_solve(pp_handle h, const double* rc_dev);                     // dense.hip
int ppi_btd_factor_schur(pp_handle h, const double* Q_host, long long corner_nnz);   // bcr.hip
int ppi_btd_coupling_solve(pp_handle h, const double* rc_dev);                       // bcr.hip
extern "C" int ppi_allreduce_sum(pp_handle h, double* buf, size_t count);            // api.hip (RCCL on the handle's stream)
int ppi_build_residual_records(pp_handle h, ppd::Group* g, const std::vector<int>& rawmap);                          // refine.hip
int ppi_residual_value_map(pp_handle h, ppd::Group* g, const std::vector<int>& ms, const std::vector<double>& mc);   // refine.hip

// Host-side helpers (internal linkage: every translation unit carries the ones it uses)
namespace {
using namespace ppd;


int build_btd_schedule(pp_handle h);

size_t schur_doubles(pp_handle h) {
  return h->btd ? (size_t)(2 * h->G - 1) * h->gs * h->gs : (size_t)h->nc * h->nc;
}

int fail(pp_handle h, int status, const std::string& msg) {
  if (h) { std::lock_guard<std::mutex> lk(h->err_mu); h->err = msg; }
  return status;
}

// The handle's stream waits for a dense factorisation of S that runs on the side stream (no-op otherwise): called by
// everything that reads the factor of S or overwrites S.
[[maybe_unused]] int join_dense(pp_handle h) {
  if (h->dense_pending) {
    h->dense_pending = false;
    if (hipStreamWaitEvent(h->stream, h->ev_dense_done, 0) != hipSuccess) return fail(h, 3, "hipStreamWaitEvent failed (dense phase)");
  }
  return 0;
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
  if (const char* e = pp::env_switch("PP_TRANSPOSE_TILES")) return std::max(1, std::atoi(e));
  return 1;
}

void free_group(Group* g) {
  for (void* p : {(void*)g->map_src, (void*)g->map_coef, (void*)g->src_own, (void*)g->fent_src, (void*)g->res_vsrc, (void*)g->res_csrc,
                  (void*)g->res_bvsrc, (void*)g->res_bcsrc}) if (p) (void)hipFree(p);
  if (g->shift_row) (void)hipFree(g->shift_row);
  if (g->shift_cls) (void)hipFree(g->shift_cls);
  for (void* p : g->value_allocs) (void)hipFree(p);
  for (void* p : g->allocs) (void)hipFree(p);
  delete g->pending;
  delete g;
}

void free_globals(pp_handle h) {
  for (void* p : {(void*)h->S_own, (void*)h->Sfac, (void*)h->Sldl, (void*)h->dvec, (void*)h->dense_mode, (void*)h->Qd, (void*)h->work, (void*)h->rs_own, (void*)h->rcd,
                  (void*)h->xc, (void*)h->ipiv, (void*)h->bkinfo, (void*)h->counters})
    if (p) (void)hipFree(p);
  h->S = h->S_own = h->Sfac = h->Sldl = h->dvec = h->Qd = h->work = h->rs = h->rs_own = h->rcd = h->xc = nullptr;
  if (h->corner_pos) (void)hipFree(h->corner_pos);
  if (h->corner_val) (void)hipFree(h->corner_val);
  h->corner_pos = nullptr; h->corner_val = nullptr; h->corner_cap = 0; h->corner_used = false; h->corner_nnz = -1;
  h->dense_mode = nullptr;
  h->ipiv = h->bkinfo = h->counters = nullptr;
  if (h->vec_part) { (void)hipFree(h->vec_part); h->vec_part = nullptr; }
  for (void* p : {(void*)h->resid_best, (void*)h->resid_ax, (void*)h->resid_rc, (void*)h->xc_save}) if (p) (void)hipFree(p);
  h->resid_best = h->resid_ax = h->resid_rc = h->xc_save = nullptr;
  h->resid_rc_valid = false; h->have_Q = false;
  if (h->resid_host) (void)hipHostFree((void*)h->resid_host);
  h->resid_host = nullptr; h->resid_dev = nullptr; h->resid_seq = 0; h->refining = false;
  if (h->dn_z) { (void)hipFree(h->dn_z); h->dn_z = nullptr; }
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
    int64_t dbl = (int64_t)g->batch * g->nraw + (int64_t)std::max(g->nraw_used, 1) * bp + 2 * (P.usize + PP_WMAX) * bp +
                  (int64_t)P.dsize * bp + (int64_t)std::max(P.n + g->nc_loc, std::max(P.bsize, 1)) * bp +
                  (int64_t)P.n * bp + 2 * (int64_t)g->batch * P.n + (int64_t)d.nchunk * std::max(std::max(g->ntiles, 1) * 64, g->nmt_items * (g->mt_wide ? 1024 : 256)) +
                  (int64_t)d.nchunk * std::max(g->nc_loc, 1) +
                  (g->cmap_host.empty() ? 0 : ((int64_t)PP_SCHUR_SLICES * std::max(g->ntiles, 1) * 64 + std::max(g->nc_loc, 1)) * bp);
    total += 8 * dbl + 2 * (int64_t)P.npiv * bp;
  }
  return total;
}

// The compare-while-staging path (api.hip: pp_stage_upload_verified_begin) sends only the pieces of a row that differ
// from the pinned staging row, trusting staged_row_valid to say that raw_own mirrors that row.  Every other writer of
// raw_own (pp_upload_values, pp_upload_sources, a caller holding pp_raw_buffer) and the release of the buffer end the mirror.
inline void invalidate_stage_mirror(Group* g) {
  std::fill(g->staged_row_valid.begin(), g->staged_row_valid.end(), (uint8_t)0);
}

void free_value_storage(Group* g) {
  invalidate_stage_mirror(g);
  for (void* p : g->value_allocs) (void)hipFree(p);
  g->value_allocs.clear();
  GroupDev& d = g->dev;
  d.raw = d.rawT = d.U = d.L = d.Dinv = d.Tm = d.Y = d.X = d.rhs = d.xout = d.Spart = d.rspart = nullptr;
  d.codes = nullptr;
  d.growth = nullptr;
  d.Sloc = d.XCL = nullptr;
  g->raw_own = g->rhs_own = g->xout_own = g->rawT_own = nullptr;
  g->res_R = g->res_D = nullptr;
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
    // (PP_WMAX zero rows behind the panels of U and of L: what a tile record names for a row or column its source lacks)
    if ((rc = value_alloc(h, g, &d.U, (size_t)(P.usize + PP_WMAX) * bp))) break;
    if ((rc = value_alloc(h, g, &d.Dinv, (size_t)P.dsize * bp))) break;
    if ((rc = value_alloc(h, g, &d.L, (size_t)(P.usize + PP_WMAX) * bp))) break;
    // the pivot-block slots of L are never written (only the rows below the block are): define them once
    if (hipMemset(d.L, 0, (size_t)(P.usize + PP_WMAX) * bp * sizeof(double)) != hipSuccess) { rc = fail(h, 3, "hipMemset failed"); break; }
    if (hipMemset(d.U + (size_t)P.usize * bp, 0, (size_t)PP_WMAX * bp * sizeof(double)) != hipSuccess) { rc = fail(h, 3, "hipMemset failed"); break; }
    if ((rc = value_alloc(h, g, &d.Y, (size_t)std::max(P.n + nc, std::max(P.bsize, 1)) * bp))) break;
    d.Tm = d.Y;     // term magnitudes of the pivot blocks (gather -> scale of one level) share the rows of the solve vector
    double* keep_x = (d.xout && d.xout != g->xout_own) ? d.xout : nullptr;
    d.xout = keep_x;
    if ((rc = value_alloc(h, g, &d.Spart, (size_t)d.nchunk * (size_t)std::max(std::max(g->ntiles, 1) * 64, g->nmt_items * (g->mt_wide ? 1024 : 256))))) break;
    if ((rc = value_alloc(h, g, &d.rspart, (size_t)d.nchunk * std::max(nc, 1)))) break;
    if (!g->cmap_host.empty()) {
      if ((rc = value_alloc(h, g, &d.Sloc, (size_t)PP_SCHUR_SLICES * std::max(g->ntiles, 1) * 64 * bp))) break;
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


static Group* get_group(pp_handle h, int group) {
  if (!h || group < 0 || group >= (int)h->groups.size()) return nullptr;
  return h->groups[group];
}

}  // namespace
