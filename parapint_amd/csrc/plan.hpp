// Symbolic plan for the batched LDL^T of one pattern group.
//
// One "group" = all scenario/time blocks K_i of this rank that share one sparsity
// pattern (and one border pattern A_i).  The plan is computed once on the host
// (reference analogue: MA27A symbolic per block, parapint/linalg/ma27_interface.py:52-92,
// plus _get_sc_structure, mpi_explicit_schur_complement.py:228-255) and drives every
// device kernel.  All instances of the group execute the same plan; the instance index
// is the SIMD lane ("lane = scenario"), so every value array is laid out
// [entry][instance] and every access is a coalesced 512-byte wave access.
//
// Augmented elimination: the block [[K_i, A_i^T], [A_i, 0]] is factorised with the
// n_c coupling rows constrained last and never eliminated, so the partial factor
// directly carries  -A_i K_i^{-1} A_i^T  (the Schur contribution the reference forms
// column by column at mpi_explicit_schur_complement.py:313-333) and
// -A_i K_i^{-1} r_i  (mpi_...:381-385).
//
// Pivots are static: 1x1 or 2x2, chosen by a minimum-degree ordering constrained so
// that a node with a (numerically) weak diagonal is only eliminated after a neighbour
// has given it a diagonal update, or inside a 2x2 pivot; sub-pivots along elimination-tree
// chains are merged into block pivots (supernodes) of up to PP_WMAX columns.  Storage is
// "L form" (see FTask below): the unscaled panel [ P_p ; U_p ] and the scaled rows
// L_p = U_p inv(P_p) with identical indexing, inv(P_p) kept separately, so all row chunks of
// a panel are independent tasks and every update is a two-operand gather.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

#include "switches.hpp"

#ifndef PP_WMAX
#define PP_WMAX 4   // widest block pivot (supernode); pivot.hpp / kernels are instantiated for bounds 1, 2, 4 (8 builds too)
#endif

// widest root front (its inertia code keeps 4 bits per count, its rows live in the registers of 16 waves)
#define PP_FRONT_MAX 15

namespace pp {

struct PlanOptions {
  int max_task_entries = 24;   // entries per gather chunk of a big panel
  int fuse_task_entries = 24;  // panels with at most this many entries are one fused task (gather + invert + scale)
  int scale_task_rows = 8;     // rows per scale task of a big panel
  double row_split_factor = 2.0;  // a row longer than this many task caps is split over the waves of one quad
  int task_order = 0;     // unsplit gather tasks of a level: 0 longest first, 1 panel by panel (rows of a panel share their L operands)
  // Top of the elimination tree ("tail"): levels holding at most tail_piv_max pivots each get their own task
  // size (few panels, long rows: shorter tasks give more waves per level).
  int tail_piv_max = 48;
  int tail_task_entries = 16;
  int tile = 8;           // register tile edge of the Schur (SYRK) kernel
  int sn_wmax = 4;        // widest supernode (columns) below the top of the tree; 1 disables merging of sub-pivots
  int sn_tol_rows = 1;    // padded rows tolerated when merging a sub-pivot into its parent
  // Top of the elimination tree: tree heights holding at most sn_tail_pop sub-pivots form dependent chains
  // (one launch pair per level, latency-bound).  There a sub-pivot is merged into its parent even if up to
  // sn_tail_tol_frac of the parent's structure has to be padded, up to sn_tail_wmax columns: a few padded
  // rows in a handful of panels buy a shorter critical path.
  // MEASURED (C3): with 4-wide panels (the default) the greedy bottom-up merge leaves the last chain of the tree as
  // 2 + 2 columns over 200 / 199 coupling rows; a tolerance of 3 % of the structure merges them: 18 -> 17 levels,
  // 722 -> 736 it/s (any sn_tail_pop in 4..32 and frac in 0.01..0.1 gives the same plan).  With a PP_WMAX = 8
  // build, relaxed merging cut 18 levels to 16 but made the factor phase slower (0.74 -> 1.03 ms at frac 0.25,
  // 1.29 ms at 1.0: 8-wide panels need 2-3x the registers in the gather / invert kernels and pad the tall
  // coupling panels).
  int sn_tail_pop = 8;
  int sn_tail_wmax = PP_WMAX;
  double sn_tail_tol_frac = 0.03;
  int md_delta_abs = 3;   // minimum-degree tolerance (absolute) for height-aware selection
  double md_delta_rel = 0.5;   // ... and relative to the current minimum degree
  double pivot_threshold = 0.01;  // 1x1 pivot accepted if |d| >= threshold * max|row| (MA27 cntl(1) analogue)
  // Elimination order: 0 = one sub-pivot at a time, lowest prospective level first within the degree window
  // (rounds 1-2 of this project); 1 = rounds of independent clusters of up to sn_wmax columns (symbolic.cpp, step 2);
  // 2 = both, the cheaper schedule kept (build_plan).
  int order_mode = 2;
  // order_mode 2: tolerance windows (md_delta_abs, md_delta_rel) planned besides the configured one -- 1: (1, 0.2),
  // 2: also (8, 1.0) -- the cheapest schedule kept (tune_for_mapped_group)
  int order_candidates = 0;
  // rounds (order_mode 1): a round with at most round_relax_pop candidates lets a sub-pivot join a cluster if it adds
  // at most max(round_relax_tol_rows, round_relax_tol_frac * |structure|) rows (else sn_tol_rows); a round with more
  // than round_narrow_pop candidates keeps its clusters at round_narrow_wmax columns
  int round_relax_pop = 600;
  int round_relax_tol_rows = 2;
  double round_relax_tol_frac = 0.05;
  // 1: pad every panel so that it holds all columns of a block pivot or none (rounds 1-2); 0: single-column entries
  int close_supernodes = 0;
  // root front (symbolic.cpp, step 2c): widest merged chain of the top of the tree (<= PP_FRONT_MAX; <= PP_WMAX: off),
  // rows of the parent a link may lack, as a fraction of the parent's rows; rows per workgroup of k_scale_wide
  int front_max = PP_FRONT_MAX;
  double front_pad_frac = 1.0;
  int front_scale_rows = 8;
  int round_narrow_pop = 1 << 30;
  int round_narrow_wmax = 2;
  // Chain fronts (symbolic.cpp, step 5b): a chain of block pivots p_1 -> p_2 -> ... in which every panel's rows are exactly
  // the columns and rows of the next one is ONE dense supernode that the <= PP_WMAX-column panels cut into dependent
  // levels (a time block of a dynamic problem: fronts of 100-170 rows eliminated four columns per level, 10-14 levels
  // each).  Such a chain of at least chain_min_panels panels, at most chain_wmax columns and chain_lds_doubles of front
  // storage is factorised in ONE level: its rows gather their contributions from outside the chain by the ordinary tasks
  // (one launch), then one workgroup per instance holds the front's pivot columns in LDS and runs the panels one after
  // the other (k_chain_front: block inversion, scaling, update of the later panels' columns).  0 disables.
  int chain_fronts = 1;
  int chain_min_panels = 3;
  int chain_min_rows = 32;           // rows of the front (a chain of tiny panels is not worth a workgroup per instance)
  int chain_wmax = 64;
  int chain_lds_doubles = 9000;      // m * (W + 1) + 4 m + ...: 72 KB -- two fronts per compute unit (MEASURED at C4: 18000, one per unit, 159 against 162 it/s)
  // Tile tasks: the rows of a chain front's panels are dense against their sources (the fronts below), so they are
  // gathered four rows at a time against whole SOURCE PANELS -- one record per (tile of 4 rows, source panel): its <= 4
  // columns are 4 + 4 operand loads for 16 multiply-adds each, against 1 + 4 loads per 4 multiply-adds and a record per
  // source column of the row tasks (k_gather_tiles).  tile_task_records: records per piece of a tile (<= PP_QUAD pieces).
  // chain fronts in the solve sweeps: the columns of a front are one level; the contributions of outside panels come
  // through the ordinary row / column tasks, the panels of the front are walked by k_chain_fwd / k_chain_bwd
  int chain_sweeps = 1;
  int batch_hint = 0;                // instances of the group (tune_for_batch)
  int chain_min_batch = 1;           // instances a pattern group needs for chain fronts (PP_PLAN_TUNE experiments)
  int chain_tiles = 1;
  int tile_task_records = 24;
  // ... and for any other panel whose rows are dense against their sources (the coupling rows of the top panels of a
  // scenario block): a panel with at least tile_min_entries product entries whose tile form requests at most
  // tile_load_ratio of the operands of its row form
  int tile_panels = 0;
  int tile_min_entries = 256;
  double tile_load_ratio = 0.7;
};

// Task sizes by batch (instances of the pattern group on this rank).  MEASURED on one MI355X (tools/tune_sweep.sh): with
// a few chunks of 64 instances a gather launch is a chain of memory round trips per wave, and a wave keeps only so many
// misses in flight -- tasks of 8 entries and rows split from 0.5 caps on spread the same entries over three times the
// waves: 128 blocks (one rank's share of C3 at 8 GPUs) 0.589 -> 0.551 ms per step, 256 blocks 0.649 -> 0.627,
// C2 (64 blocks) 0.518 -> 0.454-0.484; at 512 blocks the gain is 1 % for C3-shaped blocks and a loss of 3-4 % for the
// denser C4 / C5 blocks (8.30 -> 8.60 ms, 3.09 -> 3.21 ms), at 1024 blocks the larger tasks are as fast and hold less
// index data: the small sizes up to 256 instances.
inline void tune_for_batch(PlanOptions& o, int batch) {
  o.batch_hint = batch;
  if (batch > 0 && batch <= 256) {
    o.max_task_entries = 8;
    o.tail_task_entries = 8;
    o.row_split_factor = 0.5;
    // round 3 (after the root front and the column-run entries): panels fused up to 16 entries and scale tasks of 4
    // rows -- 128 blocks 0.452 -> 0.434 ms, 64 blocks 0.425 -> 0.417, 256 blocks 0.513 -> 0.496, C2 0.403 -> 0.387
    // (the same sizes change nothing at 1024 blocks: 0.433-0.437 ms of factor levels for fuse 16..32, scale rows 4..8)
    o.fuse_task_entries = 16;
    o.scale_task_rows = 4;
  }
}

// Pattern groups with coupling rows of their own per instance (pp_add_group_mapped: the time blocks of a dynamic problem) are
// deep chains -- 60 ... 100 levels -- whose depth and fill depend on the tolerance window of the minimum-degree order far more
// than those of a scenario block: the KKT block of the time-staged quadratic program of examples/dynamics_qp.py orders into
// 96 levels / 150 k factor entries with the default window (3, 0.5), 68 / 127 k with (1, 0.2) and 83 / 155 k with (8, 1.0), while
// the C3 scenario block goes from 11 levels to 209 with (1, 0.2).  For a mapped group of more than one chunk the two other
// windows are planned besides (build_plan: all of them side by side on host threads) and the cheapest schedule is kept; the
// Burgers and the synthetic C4 blocks keep the default (63 ... 66 levels under all three).
inline void tune_for_mapped_group(PlanOptions& o, int batch) {
  if (batch > 64) {
    static const char* e = pp::env_switch("PP_ORDER_CANDIDATES");
    o.order_candidates = e ? std::atoi(e) : 2;
  }
}

// Developer knob for schedule experiments (environment PP_PLAN_TUNE = "max_task_entries=48,scale_task_rows=16,..."),
// read by the library and by the test interpreter alike.  Returns false and names the key if one is unknown.
inline bool apply_plan_tune(PlanOptions& opt, const char* tune, std::string& bad_key) {
  if (!tune) return true;
  std::string t(tune);
  size_t pos = 0;
  while (pos < t.size()) {
    size_t end = t.find(',', pos);
    if (end == std::string::npos) end = t.size();
    const std::string kv = t.substr(pos, end - pos);
    const size_t eq = kv.find('=');
    if (eq != std::string::npos) {
      const std::string k = kv.substr(0, eq);
      const double v = std::atof(kv.c_str() + eq + 1);
      if (k == "max_task_entries") opt.max_task_entries = (int)v;
      else if (k == "fuse_task_entries") opt.fuse_task_entries = (int)v;
      else if (k == "scale_task_rows") opt.scale_task_rows = (int)v;
      else if (k == "tail_task_entries") opt.tail_task_entries = (int)v;
      else if (k == "tail_piv_max") opt.tail_piv_max = (int)v;
      else if (k == "sn_tail_pop") opt.sn_tail_pop = (int)v;
      else if (k == "sn_tail_wmax") opt.sn_tail_wmax = (int)v;
      else if (k == "sn_tail_tol_frac") opt.sn_tail_tol_frac = v;
      else if (k == "sn_wmax") opt.sn_wmax = (int)v;
      else if (k == "sn_tol_rows") opt.sn_tol_rows = (int)v;
      else if (k == "md_delta_abs") opt.md_delta_abs = (int)v;
      else if (k == "md_delta_rel") opt.md_delta_rel = v;
      else if (k == "row_split_factor") opt.row_split_factor = v;
      else if (k == "task_order") opt.task_order = (int)v;
      else if (k == "order_candidates") opt.order_candidates = (int)v;
      else if (k == "order_mode") opt.order_mode = (int)v;
      else if (k == "pivot_threshold") opt.pivot_threshold = v;
      else if (k == "front_max") opt.front_max = std::min((int)v, (int)PP_FRONT_MAX);   // (k_front_invert / k_scale_wide hold at most PP_FRONT_MAX columns)
      else if (k == "front_pad_frac") opt.front_pad_frac = v;
      else if (k == "front_scale_rows") opt.front_scale_rows = (int)v;
      else if (k == "close_supernodes") opt.close_supernodes = (int)v;
      else if (k == "round_relax_pop") opt.round_relax_pop = (int)v;
      else if (k == "round_relax_tol_rows") opt.round_relax_tol_rows = (int)v;
      else if (k == "round_relax_tol_frac") opt.round_relax_tol_frac = v;
      else if (k == "round_narrow_pop") opt.round_narrow_pop = (int)v;
      else if (k == "round_narrow_wmax") opt.round_narrow_wmax = (int)v;
      else if (k == "chain_fronts") opt.chain_fronts = (int)v;
      else if (k == "chain_min_panels") opt.chain_min_panels = (int)v;
      else if (k == "chain_min_rows") opt.chain_min_rows = (int)v;
      else if (k == "chain_wmax") opt.chain_wmax = (int)v;
      else if (k == "chain_lds_doubles") opt.chain_lds_doubles = (int)v;
      else if (k == "chain_tiles") opt.chain_tiles = (int)v;
      else if (k == "chain_sweeps") opt.chain_sweeps = (int)v;
      else if (k == "tile_panels") opt.tile_panels = (int)v;
      else if (k == "tile_min_entries") opt.tile_min_entries = (int)v;
      else if (k == "tile_load_ratio") opt.tile_load_ratio = v;
      else if (k == "chain_min_batch") opt.chain_min_batch = (int)v;
      else if (k == "tile_task_records") opt.tile_task_records = (int)v;
      else { bad_key = k; return false; }
    }
    pos = end + 1;
  }
  return true;
}


// Factor schedule ("L form").  For every block pivot p two panels are stored with the same
// indexing: the unscaled panel U_p = [P_p ; U_p] and the scaled rows L_p = U_p inv(P_p).  The
// update of a destination ROW (all w columns of the block pivot at once) is a pure gather over
// row entries e:
//     acc[q0 + j] -= U[e.u] * L[e.l + j * e.wk]   for j < m  (source column t of panel k: U_k[i][t], L_k[p_(q0+j)][t];
//                                                             e.q = q0 | m << 4: a run of m consecutive columns of the
//                                                             block pivot -- all w of them when panel k holds the whole
//                                                             block pivot as rows, fewer when it holds only some)
//     acc[e.q] += input value ~e.u                           (initial-value entry: e.u < 0, e.l < 0)
// so every U operand is loaded once per row, no per-task multiplier tables are needed and block
// pivots of any width cost (1 + w) loads per w multiply-adds.
// Task kinds:  0 gather chunk of a big panel (stores U, and the term magnitudes of pivot-block
//                scalars for the zero-pivot test),
//              1 fused small panel (gather everything, invert the block, scale the rows, store U, L,
//                inv(P), inertia code),
//              2 scale chunk of a big panel (inverts the gathered block, L rows = U rows inv(P); the
//                chunk with r0 == w also stores inv(P) and the inertia code).
// Per level: one launch of the kind-0/1 tasks, then (if any) one launch of the kind-2 tasks.
// Gather / fused tasks are executed by workgroups of PP_QUAD waves, one task per wave ("quad"); the task list of a
// level is a whole number of quads (padded with kind -1 = no-op).  A row whose entry list is long -- the pivot
// rows of the top panels collect an update from most of their subtree, 50-80 entries against a dozen for the
// other rows, and the dependent round trips of the longest task set the duration of a launch -- is cut into
// npieces <= PP_QUAD pieces that fill ONE quad: every wave gathers its piece, the partial sums meet in LDS and
// the wave of piece 0 adds them in piece order (deterministic) and stores the row.
#ifndef PP_QUAD
#define PP_QUAD 4
#endif

// qoff / ws: column slice of the panel the task gathers (ws == w, qoff == 0 except in the root front, whose w > PP_WMAX
// columns are gathered PP_WMAX at a time).  kind 3: scale chunk of the root front (k_scale_wide).
// kind 4: tile task (PlanOptions::chain_tiles): rows [r0, r1) (<= PP_TILE_ROWS) of a panel against the source panels
// trec[te0 .. te1) (device form); its per-row entry lists (dptr0, as for kind 0: the initial values in piece 0, the
// product entries of this piece's sources) are what the host interpreter runs and where the device takes the initial values.
#define PP_TILE_ROWS 4
#define PP_TILE_DEPTH 2      // steps (source columns) in flight per wave in k_gather_tiles
#define PP_TREC_INTS 12
struct FTask { int piv, r0, r1, dptr0, kind, piece = 0, npieces = 1, qoff = 0, ws = 0, te0 = 0, te1 = 0; };
struct FEntry { int u, l, wk, q; };
// Schur tile record: pivot p contributes to tile (ta, tb); slots (or -1) of the tile's
// coupling rows inside panel p
struct STileRec { int piv; int slotA[8]; int slotB[8]; };

struct Plan {
  int n = 0, nc = 0, npiv = 0, ncan = 0;
  int n_levels = 0;
  PlanOptions opt;
  std::vector<int> perm, iperm;          // new->old, old->new (K nodes)
  std::vector<int> piv_start, piv_w;     // first new column of block pivot (supernode) p, width 1..PP_WMAX
  std::vector<unsigned> piv_sub;         // bit i: columns i, i+1 of the block form a 2x2 sub-pivot
  std::vector<int> piv_cslot0, piv_ncrow;  // first coupling-row slot of the panel, number of coupling rows
  std::vector<int> piv_of_col;           // new column -> pivot
  std::vector<int> piv_rowptr, rowidx;   // rows below pivot p (new indices; n+c = coupling row c)
  std::vector<int64_t> piv_uoff;         // panel offset, doubles per lane
  int64_t usize = 0;                     // doubles per lane in U storage
  std::vector<int> piv_doff;             // offset of inv(P_p) in Dinv storage: w(w+1)/2 scalars, lower triangle by rows
  int dsize = 0;                         // doubles per lane in Dinv storage
  std::vector<int64_t> pos_of_can;       // canonical input entry -> U position
  std::vector<int> piv_level;            // etree height of pivot
  // factor schedule
  std::vector<FTask> ftasks;             // gather / fused tasks sorted by level
  std::vector<FTask> stasks;             // scale tasks sorted by level
  std::vector<int> fdst_ptr;             // per gather/fused task (nrows + 1) offsets into fentries, at dptr0
  std::vector<FEntry> fentries;
  std::vector<int> flevel_ptr;           // n_levels+1 -> ftasks
  std::vector<int> slevel_ptr;           // n_levels+1 -> stasks
  std::vector<int> piv_boff;             // offset of the pivot block's w*w term magnitudes (Tm storage)
  int bsize = 0;
  std::vector<int> flevel_maxent;        // per level: max entries of a gather/fused task (a piece counts as its whole row)
  std::vector<int> flevel_nsplit;        // per level: number of split rows (0: the level runs one wave per workgroup)
  std::vector<int> clevel_ptr, clevel_col;  // solve schedule: scalar columns (new indices) by level
  std::vector<int> clevel_nchain;        // per level: columns of chain fronts (the last ones of the level's list)
  bool chain_sweeps_on = false;          // the sweeps use the factor levels and the chain kernels (PlanOptions::chain_sweeps)
  int tail_level0 = 0;                   // levels >= tail_level0 form the tail (== n_levels: no tail)
  // Root front: the last block pivot when it is wider than PP_WMAX (else -1).  It is alone on the last level; its
  // gather tasks (column slices) are in ftasks like all others, its pivot block is inverted by k_front_invert and its
  // rows scaled by k_scale_wide in chunks wtasks (kind 3).
  int front_piv = -1;
  std::vector<FTask> wtasks;
  // Chain fronts (PlanOptions::chain_fronts): front c = panels chain_piv[chain_ptr[c] .. chain_ptr[c + 1]) in elimination
  // order, chain_col0[.] = first column of each inside the front; chain_m / chain_w = rows (pivot columns included) and
  // pivot columns of the front; chain_level = the factor level at which its rows are gathered and the front is run.
  // piv_chain[p] = front of pivot p or -1.  piv_flevel[p] = level of p in the FACTOR schedule (== piv_level[p] for a plan
  // without chain fronts; all panels of a chain share one); piv_level keeps the dependency levels the solve sweeps use.
  std::vector<int> chain_ptr, chain_piv, chain_col0, chain_m, chain_w, chain_level, piv_chain, piv_flevel;
  std::vector<int> chain_lvl_ptr;        // n_levels + 1 -> fronts (sorted by level)
  // tile tasks (kind 4), sorted by level in whole quads like ftasks; records {w_k, U position of row i (k = 0) x 4, L position
  // of column q x 4, 0, 0, 0}: positions >= usize name the PP_WMAX zero rows behind the panels (a row or column the source lacks)
  std::vector<FTask> ttasks;
  std::vector<int> tlevel_ptr;           // n_levels + 1 -> ttasks
  std::vector<int> trec;
  int n_flevels = 0;                     // levels of the factor schedule that hold anything
  // forward-solve entries: scalar row (new column index c) = b_c - sum U[upos] * z[zcol]
  std::vector<int> sfwd_eptr;            // n+1 -> sfwd_upos / sfwd_zcol
  std::vector<int> sfwd_upos, sfwd_zcol;
  std::vector<int> crow_eptr, crow_upos, crow_zcol;  // same for the coupling rows
  // solve schedule
  std::vector<int> lvl_ptr, lvl_piv;     // pivots by level
  // Schur (SYRK) schedule
  std::vector<int> stile_a, stile_b, stile_ptr;  // tiles (ta>=tb) and record ranges
  std::vector<STileRec> stile_rec;
  // statistics
  int64_t nnz_L = 0;        // structural entries of L below the pivot blocks
  int64_t flops_factor = 0; // multiply-adds of the panel updates
  int64_t flops_schur = 0;  // multiply-adds of the coupling x coupling update
  int n_2x2 = 0;
  bool numeric_ordering = false;  // ordered with representative values (else pattern only)
  std::string error;
};

// K pattern: unique lower-triangular entries (row >= col), any order; entry e is
// canonical value e.  Border pattern: entries (coupling row, block column) are canonical
// values nnzK + e.  vals[ncan]: canonical values of a representative instance used to fix
// the static pivot sequence (may be null: pattern-only ordering, every entry taken as 1 and
// absent diagonals as zero).  Returns 0 on success.
int build_plan(int n, int nc, int nnzK, const int* rowK, const int* colK,
               int nnzB, const int* rowB, const int* colB, const double* vals,
               const PlanOptions& opt, Plan& plan);

}  // namespace pp
