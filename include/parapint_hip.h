/* parapint_hip.h -- C ABI of the MI355X-native Schur-complement KKT solver.
 *
 * Drop-in boundary for the hot path of sandialabs/parapint:
 *   parapint.linalg.MPISchurComplementLinearSolver  (parapint/linalg/schur_complement/
 *   mpi_explicit_schur_complement.py:128-452) together with its per-block sub-solvers
 *   (ma27_interface.py:9-256, mumps_interface.py:11-229, scipy_interface.py:11-67).
 * The reference has no FFI of its own (it is pure Python over third-party solvers); this
 * header is what a ctypes binding for that path binds.  Plain pointers and
 * sizes only; every function returns a LinearSolverStatus value
 * (parapint/linalg/results.py:4-9): 0 successful, 1 not_enough_memory, 2 singular, 3 error,
 * 4 warning.  Nothing throws across the ABI; pp_last_error() gives the message.
 *
 * Life cycle (one handle per solver object, bound to one HIP device and one stream):
 *   pp_create -> pp_begin_symbolic -> pp_add_group* -> pp_end_symbolic
 *             -> { pp_upload_values* -> pp_numeric_local -> [all-reduce of pp_schur_buffer]
 *                  -> pp_factor_schur -> pp_get_status
 *                  -> { pp_upload_rhs* -> pp_solve_forward -> [all-reduce of pp_rs_buffer]
 *                       -> pp_solve_coupling -> pp_solve_backward -> pp_download_solution* }* }*
 *   -> pp_destroy
 * The two all-reduces (reference: comm.Allreduce at mpi_...:343 and :387) are issued by the
 * caller (RCCL through torch.distributed) on the device buffers returned here.
 *
 * A "group" is the set of this rank's diagonal blocks K_i that share one sparsity pattern and
 * one border pattern A_i; all its instances are factorised together, one instance per SIMD lane.
 */
#ifndef PARAPINT_HIP_H
#define PARAPINT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pp_solver* pp_handle;

/* Creates a handle on HIP device `device` (-1: current).  `stream` is a hipStream_t (or NULL for
 * the default stream) on which every kernel and copy of this handle is enqueued.
 * Replaces: MPISchurComplementLinearSolver.__init__ (mpi_...:154-163). */
int pp_create(pp_handle* out, int device, void* stream);
void pp_destroy(pp_handle h);
const char* pp_last_error(pp_handle h);
/* SHA-1 of the kernel sources (parapint_amd/csrc) this library was built from: the loader refuses a library whose stamp
 * differs from the sources beside it (a stale build).  No reference counterpart (a build-system guard). */
const char* pp_source_sha1(void);

/* ---- symbolic phase: do_symbolic_factorization (mpi_...:165-255) ------------------------- */
int pp_begin_symbolic(pp_handle h, int n_coupling);

/* One pattern group.
 *   n, batch            block dimension and number of local blocks with this pattern
 *   rowK/colK[nnzK]     canonical pattern of tril(K_i): unique entries, row >= col
 *   rowB/colB[nnzB]     canonical pattern of A_i: (coupling row, block column), unique
 *   nraw                length of the raw value vector the caller supplies per block
 *                       (K_i's COO data followed by A_i's COO data, in the caller's order,
 *                       duplicates and upper-triangle entries allowed -- quirk Q7 of SURVEY.md)
 *   can_ptr/can_idx     CSR map canonical entry e (0..nnzK+nnzB-1) -> raw positions summed into it
 *   rep_vals[nnzK+nnzB] canonical values of a representative block used to fix the static pivot
 *                       sequence (MA27A+MA27B pivot-choice analogue), or NULL (pattern only)
 * Replaces: per-block sub-solver do_symbolic_factorization (ma27_interface.py:52-92) and
 * _BorderMatrix (mpi_...:33-58). */
int pp_add_group(pp_handle h, int n, int batch, int nnzK, const int32_t* rowK, const int32_t* colK,
                 int nnzB, const int32_t* rowB, const int32_t* colB, int nraw, const int32_t* can_ptr,
                 const int32_t* can_idx, const double* rep_vals, int* group_out);

/* A group whose instances touch coupling rows of their own (time blocks of a dynamic problem: block t is linked only to
 * the coupling variables of its two neighbours, sc_ip_interface.py:308-333; scenarios that see a subset of the
 * first-stage variables): rowB holds LOCAL coupling rows 0 .. nc_loc-1 and cmap[batch][nc_loc] gives, per instance, the
 * global coupling index of each.  The instances are still factorised together, one per lane; their Schur cliques,
 * coupling right-hand sides and coupling solutions are scattered / gathered through the map (what the reference does
 * with sc_data_slices, mpi_...:228-255, 313-333).  cmap = NULL, nc_loc = n_c is pp_add_group. */
int pp_add_group_mapped(pp_handle h, int n, int batch, int nnzK, const int32_t* rowK, const int32_t* colK, int nnzB,
                        const int32_t* rowB, const int32_t* colB, int nraw, const int32_t* can_ptr, const int32_t* can_idx,
                        const double* rep_vals, int nc_loc, const int32_t* cmap, int* group_out);

/* Structure of S, right after pp_begin_symbolic.  mode 0 (default): dense n_c x n_c (the union of the border cliques of a
 * stochastic program is dense).  mode 1: block-tridiagonal with G diagonal blocks of gs rows (n_c = G * gs; the caller
 * orders and pads the coupling variables so that every clique and every entry of Q lies in one block or two adjacent
 * ones -- the banded S of a time-staged problem, mpi_...:88-125).  The Schur buffer is then D[G][gs][gs] | E[G-1][gs][gs]
 * (E_t = S(block t+1, block t), column-major inside a block) followed by the same 8-double tail, Q is handed over in the
 * same layout, and S is factorised by a block LDL^T with Bunch-Kaufman inside the blocks; all groups must be mapped.
 * pp_schur_buffer_doubles: length of the Schur buffer (tail included) for the current structure. */
int pp_set_coupling_structure(pp_handle h, int mode, int gs, int G);
/* Elimination order of a block-tridiagonal S: 0 (default) block cyclic reduction -- log2 G levels of batched block
 * operations --, 1 ascending block order (G dependent steps; the fallback when an odd-even diagonal block is singular:
 * an indefinite S has singular principal submatrices).  May be changed between factorisations. */
int pp_set_coupling_schedule(pp_handle h, int sequential);
int64_t pp_schur_buffer_doubles(pp_handle h);

/* Builds the plans' device images and allocates all device memory (_get_sc_structure,
 * mpi_...:228-255: the dense S buffer replaces the sparse COO pattern + sc_data_slices). */
int pp_end_symbolic(pp_handle h);

/* ---- numeric phase: do_numeric_factorization (mpi_...:257-361) --------------------------- */
/* Raw values of all instances of a group, [batch][nraw] row-major.  on_device != 0: `raw` is a
 * device pointer (device-to-device copy); else host memory (async H2D on the handle's stream). */
int pp_upload_values(pp_handle h, int group, const double* raw, int on_device);
/* Device pointer of the group's raw value buffer ([batch][nraw]) for producers that assemble
 * values on the device (no copy needed before pp_numeric_local). */
double* pp_raw_buffer(pp_handle h, int group);

/* Zero-copy alternative: make the kernels read the group's raw values directly from a caller-owned
 * device buffer ([batch][nraw]); NULL restores the library's own buffer.  The buffer must stay
 * valid until pp_numeric_local has completed on the stream. */
int pp_bind_raw_buffer(pp_handle h, int group, double* dev_ptr);

/* The same for callers that hold the values on the host: only the raw entries some canonical entry reads
 * (pp_used_raw_entries: ascending raw indices, pp_group_stats out[15] raw entries of which the used ones are counted by
 * the `capacity` the call needs) are uploaded, [batch][n_used] row-major, rows [row0, row0 + nrows).  A KKT block
 * handed over with both triangles (interface.py:470-494 sets both Jacobian transposes) has 45 % of its entries unread. */
int pp_upload_values_compact(pp_handle h, int group, const double* compact, int row0, int nrows, int on_device);
int pp_used_raw_entries(pp_handle h, int group, int32_t* out, int capacity);

/* ---- f2: KKT values straight from the producer's arrays -------------------------------------------
 * The interior-point interface changes only a few arrays per iteration -- Hessian values, Jacobian values, the
 * barrier diagonals z/(x-l) + z/(u-x) (interfaces/interface.py:432-494, sc_ip_interface.py:1677-1681) -- and every
 * COO entry of K_i / A_i is one of them times +-1, or a constant.  pp_set_value_map (after pp_end_symbolic) fixes
 * that relation per raw entry e: value(e) = coef_of_raw[e] * source[src_of_raw[e]]  (src_of_raw[e] < 0: the constant
 * coef_of_raw[e]).  The sources of all instances live in ONE device buffer [nsrc][bpad] (instance index fastest,
 * bpad = batch rounded up to 64 = pp_group_stats out[2] rounded): pp_source_buffer returns the library's own,
 * pp_bind_source_buffer makes the kernels read a caller-owned one (NULL: the library's), pp_upload_sources fills the
 * library's from a [batch][nsrc] array (host or device).  After any of the last two the next pp_numeric_factor_blocks
 * gathers its input from the sources: no COO assembly, no host staging, no transposition. */
int pp_set_value_map(pp_handle h, int group, int nsrc, const int32_t* src_of_raw, const double* coef_of_raw);
double* pp_source_buffer(pp_handle h, int group);
int pp_bind_source_buffer(pp_handle h, int group, double* dev_ptr);
int pp_upload_sources(pp_handle h, int group, const double* src, int on_device);

/* Batched block factorisation + local Schur contribution: K_i = L D L^T for every local block
 * (mpi_...:292-299) and S_local = -sum_i A_i K_i^{-1} A_i^T (mpi_...:312-333).  Result is left
 * in the Schur buffer; block inertia and the singular-pivot count ride in its 4-double tail. */
int pp_numeric_local(pp_handle h);
/* The same in two calls, so that a caller can bracket the reference's timer labels separately
 * (mpi_...:291-333: 'form SC/factorize' = the block factorisations; 'form SC/back solve' + 'dot product' = the
 * Schur contributions): pp_numeric_local == pp_numeric_factor_blocks followed by pp_numeric_schur. */
int pp_numeric_factor_blocks(pp_handle h);
int pp_numeric_schur(pp_handle h);
/* The same with the Schur update -- and the dense factorisation of S that follows it -- on a stream of the library's own,
 * forked behind the factor levels (side_stream != 0; dense S, at most two pattern groups): a forward sweep enqueued on the
 * handle's stream after pp_factor_schur then runs beside both.  An all-reduce of S between the two must be the library's
 * (pp_allreduce_schur); pp_solve_coupling and everything else that reads the factor of S or writes S joins. */
int pp_numeric_schur_ex(pp_handle h, int side_stream);
/* Status agreement without a collective of its own (mpi_...:19-30, 294-305: the reference gathers the sub-solver
 * statuses of all ranks before it communicates S).  A rank whose block phase failed on the host side (status 1, 2 or 3)
 * calls this instead of pp_numeric_schur: its Schur buffer becomes a zero contribution whose tail carries the failure,
 * the rank still takes part in the all-reduce, and pp_get_status reports the most severe status on every rank. */
int pp_fail_local(pp_handle h, int status);

/* Device buffer of n_c*n_c + 8 doubles: dense column-major S_local followed by
 * {n_zero_pivots, n_pos, n_neg, host failures (pp_fail_local), instances with element growth, 3 reserved} as doubles, so ONE sum all-reduce carries the Schur
 * complement (mpi_...:343), the status agreement (mpi_...:19-30) and the inertia sums
 * (mpi_...:427-429).  pp_bind_schur_buffer lets the caller supply that memory (e.g. a torch tensor). */
double* pp_schur_buffer(pp_handle h);
int pp_bind_schur_buffer(pp_handle h, double* dev_ptr);

/* S = S_allreduced + Q, dense Bunch-Kaufman LDL^T of S, inertia(S)  (mpi_...:347-361).
 * Q: dense column-major n_c x n_c on the host (lower triangle read), or NULL for Q = 0. */
int pp_factor_schur(pp_handle h, const double* Q_host);
/* The same for a block-tridiagonal S (time-staged problems: the reference keeps S and the coupling block sparse,
 * mpi_...:88-125, 228-255, sc_ip_interface.py:308-357), Q as nnz (position in the layout of the Schur buffer, value)
 * pairs on the host; duplicates add.  The pairs are uploaded on a stream of their own -- the call does not wait for the
 * block factorisation still running on the handle's stream -- and the arrays may be reused when it returns. */
int pp_factor_schur_corner(pp_handle h, int64_t nnz, const int64_t* pos, const double* val);

/* Block pivots (supernodes) for groups added afterwards: sub-pivot chains of the elimination tree
 * are merged up to `wmax` columns (1 = off, at most 4) when at most `tol_rows` padded rows result;
 * 0 / -1 keep the built-in defaults.  Fewer, fatter levels in every sweep. */
int pp_set_supernodes(pp_handle h, int wmax, int tol_rows);

/* Number of instance groups whose level sweeps run on separate HIP streams.  0 = default (one
 * group: on MI355X / ROCm 7.2 more groups were measured slower, the launches serialise). */
int pp_set_instance_splits(pp_handle h, int nsplit);

/* Dense policy for S: 0 (default) = blocked LDL^T without pivoting on the fp64 matrix cores,
 * accepted only when all pivots share one sign (S definite), with the Bunch-Kaufman kernel as the
 * on-device fallback; 1 = Bunch-Kaufman only.  pp_get_dense_mode reports which factor the last
 * pp_factor_schur produced (1 = blocked LDL^T accepted, 0 = Bunch-Kaufman). */
int pp_set_dense_policy(pp_handle h, int policy);
int pp_get_dense_mode(pp_handle h, int* mode_out);

/* Synchronises the stream and returns {status, pos, neg, zero} of the whole matrix:
 * sum of block inertias (all ranks, taken from the all-reduced tail) + inertia(S)
 * (get_inertia, mpi_...:404-436).  status is 2 (singular) if any pivot was numerically zero. */
int pp_get_status(pp_handle h, int64_t out[4]);

/* Copies the all-reduced Schur complement (without Q), dense column-major n_c x n_c, to the host. */
int pp_get_schur(pp_handle h, double* S_host);

/* ---- back-solve: do_back_solve (mpi_...:363-402) ----------------------------------------- */
/* Right-hand sides of a group's blocks, [batch][n] row-major. */
int pp_upload_rhs(pp_handle h, int group, const double* rhs, int on_device);
double* pp_rhs_buffer(pp_handle h, int group);
/* Zero-copy alternative for right-hand sides already resident on the device (NULL restores). */
int pp_bind_rhs_buffer(pp_handle h, int group, double* dev_ptr);
/* Forward elimination of all local blocks; leaves r_s_local = -sum_i A_i K_i^{-1} r_i
 * (mpi_...:381-385) in the rs buffer (n_c doubles, device) for the caller's all-reduce (:387). */
int pp_solve_forward(pp_handle h);
/* The same with a promise: rhs_before_factor != 0 states that the bound right-hand side (pp_bind_native_vectors) was
 * complete before the block factorisation of this step was enqueued -- the situation of an interior-point iteration
 * (interior_point.py:553-566: the right-hand side exists before the matrix is factorised).  With several pattern groups
 * and a block-tridiagonal S the sweep of each group is then ordered behind the factorisation of that group only and
 * overlaps the Schur update and the factorisation of S.  0: as pp_solve_forward. */
int pp_solve_forward_ex(pp_handle h, int rhs_before_factor);
double* pp_rs_buffer(pp_handle h);
int pp_bind_rs_buffer(pp_handle h, double* dev_ptr);
/* x_c = S^{-1} (r_c + r_s)  (mpi_...:388-391); r_c on the host (n_c doubles, or NULL = 0). */
int pp_solve_coupling(pp_handle h, const double* rc_host);
/* Native vectors: right-hand sides and solutions of a group as [n][bpad] device arrays (row = the caller's row of the
 * block, instance index fastest, bpad = batch rounded up to 64) -- the layout every kernel here works in.  The forward
 * sweep then reads b where the caller keeps it (no transposition, no copy: columns without incoming entries are never
 * materialised) and the backward sweep writes x in the caller's row order.  (NULL, NULL) restores the [batch][n]
 * buffers of pp_upload_rhs / pp_download_solution; (rhs, NULL) is enough for pp_solve_forward -- a forward sweep enqueued
 * right behind pp_factor_schur overlaps the dense factorisation of S, which the library runs on a stream of its own (the
 * sweep does not depend on S; pp_solve_coupling joins) -- the solution buffer is bound before pp_solve_backward. */
int pp_bind_native_vectors(pp_handle h, int group, const double* rhs_dev, double* x_dev);
/* The same with r_c resident on the device (NULL = 0), and the device address of x_c (n_c doubles). */
int pp_solve_coupling_dev(pp_handle h, const double* rc_dev);
double* pp_coupling_solution_buffer(pp_handle h);
int pp_copy_coupling_solution(pp_handle h, double* dev_ptr);   /* x_c -> caller's device buffer, stream-ordered */
/* x_i = K_i^{-1} (r_i - A_i^T x_c) for all local blocks (mpi_...:393-396), by back substitution. */
int pp_solve_backward(pp_handle h);
/* Solutions of a group's blocks, [batch][n] row-major (host, or device if on_device). */
int pp_download_solution(pp_handle h, int group, double* x, int on_device);
double* pp_solution_buffer(pp_handle h, int group);
/* Makes pp_solve_backward write a group's solutions ([batch][n]) into a caller-owned device buffer (NULL restores). */
int pp_bind_solution_buffer(pp_handle h, int group, double* dev_ptr);
int pp_get_coupling_solution(pp_handle h, double* xc_host);

/* ---- a-posteriori check of a back-solve, iterative refinement ------------------------------------------------------------
 * The reference's sub-solvers pivot every block on its own values (MA27 with cntl(1), ma27_interface.py:36-47, 110-140;
 * SuperLU, scipy_interface.py:26-31): a `successful` factorisation of theirs solves accurately.  Here one static pivot
 * sequence serves all instances of a pattern group, so every back-solve (mpi_...:363-402) is looked at.
 *   pp_residual         after pp_solve_backward: r_i = b_i - K_i x_i - A_i^T x_c of every local block from the values the last
 *                       factorisation read, rho_b = max |r| / max (|K||x| + |A^T x_c| + |b|) over the rows of an instance,
 *                       and the local sums sum_i A_i x_i, sum_i |A_i||x_i| of the coupling rows; store != 0 keeps r as the
 *                       right-hand side of a correction solve.  bc_dev: the coupling right-hand side of the back-solve on
 *                       the device (n_c doubles) or NULL.  coupling_on_device != 0 (one rank, dense S): the coupling rows
 *                       b_c - sum A x - Q x_c (Q as the last pp_factor_schur got it) are judged on the device as well, and
 *                       with store their residual stays there for pp_refine_solve_coupling.  coupling_on_device == 2:
 *                       several ranks with the library's communicator (pp_comm_init) -- the sums of the coupling rows and one
 *                       slot per rank for its block result meet in ONE all-reduce on the handle's stream, every rank
 *                       finishes identically (collective).  Stream-ordered; publishes to a pinned mailbox.
 *   pp_residual_result  waits for it: out = {worst rho of the local instances, its group, its slot, the largest row scale
 *                       |K||x| + |A^T x_c| + |b| of the blocks, rho of the coupling rows or -1, worst rho of the instances of
 *                       ALL ranks (coupling_on_device == 2; else out[0])}; -1: coupling_out
 *                       (4 n_c doubles) receives x_c | sum A x | sum |A||x| | b_c -- the caller adds Q x_c and, with several
 *                       ranks, all-reduces the two sums before it judges the coupling rows.
 *   pp_refine_begin / pp_refine_end   bracket a correction solve: between them pp_solve_forward, the all-reduce of r_s,
 *                       pp_solve_coupling(_dev) with the residual of the coupling rows, pp_solve_backward run on the stored
 *                       residual and produce a correction; end adds it to the solution of the back-solve (x_c included) in
 *                       whatever vectors that solve used and restores them. */
int pp_residual(pp_handle h, int store, const double* bc_dev, int coupling_on_device);
int pp_residual_result(pp_handle h, double out[6], double* coupling_out);
int pp_refine_solve_coupling(pp_handle h);
int pp_refine_begin(pp_handle h);
int pp_refine_end(pp_handle h);

/* ---- misc -------------------------------------------------------------------------------- */
/* Memory reallocation protocol (interior_point.py:634-652 try_factorization_and_reallocation; the sub-solver side is
 * ma27_interface.py:126-131 status -3/-4 -> not_enough_memory and :153-154 iw_factor, a_factor *= factor; reference
 * test linalg/tests/test_realloc.py:10-61).  The device value storage (factor panels, work vectors) is sized exactly
 * by the symbolic phase.  pp_set_memory_budget caps it (bytes; 0 = no cap, the default): when the plan needs more
 * than budget x (product of the factors given to pp_increase_memory_allocation since), pp_end_symbolic still succeeds
 * and pp_upload_values / pp_numeric_local return 1 (not_enough_memory) without allocating anything, until the caller
 * has raised the budget.  A real hipErrorOutOfMemory maps to the same status and leaves nothing allocated, so a retry
 * after memory has been freed elsewhere works the same way.
 * pp_memory_info: out = {bytes the plan needs at most, effective budget (0 = none), bytes allocated now (0: nothing yet)}:
 * the [instance][entry] input copy, its transposed form and the [instance][row] copies of right-hand side and solution
 * are allocated at first use by the input / output forms that need them (host values, host vectors). */
int pp_increase_memory_allocation(pp_handle h, double factor);
int pp_set_memory_budget(pp_handle h, int64_t bytes);
int pp_memory_info(pp_handle h, int64_t out[3]);
/* Blocks until the handle's stream is idle. */
int pp_synchronize(pp_handle h);
/* Block-tridiagonal S (time-staged problems; the reference hands the sparse S to its sub-solver, mpi_...:352-361, whose
 * pivoting is MA27's threshold test, ma27_interface.py:36-47): after pp_factor_schur, out = {diagonal blocks of the cyclic
 * reduction inverted from an unpivoted LDL^T whose multipliers all passed the threshold test, blocks left to
 * Bunch-Kaufman}.  Diagnostic (synchronises the stream); {0, 0} when S is dense. */
int pp_bcr_block_paths(pp_handle h, int32_t out[2]);

/* Phase timing with HIP events on the handle's stream (measurement only, SURVEY.md section 5.1
 * timer labels): pp_profile(h, 1) resets and enables, pp_phase_times returns accumulated
 * milliseconds, kernel launches and bracket counts for the phases
 *   0 assemble (transpose + scatter)      'form SC/factorize' input
 *   1 block factorisation levels          'form SC/factorize'
 *   2 inertia count + Schur tiles/reduce  'form SC/back solve + dot product'
 *   3 dense S: add Q + Bunch-Kaufman      'factor SC'
 *   4 forward substitution levels         'back_solve'
 *   5 coupling rows of the forward sweep  'back_solve'
 *   6 coupling solve with the S factor    'back_solve'
 *   7 back substitution levels            'back_solve' */
int pp_profile(pp_handle h, int enable);
int pp_phase_times(pp_handle h, double ms_out[8], int32_t launches_out[8], int32_t calls_out[8]);

/* Per-group plan statistics.  out[16]:
 *  0 n, 1 n_coupling, 2 batch, 3 n_pivots, 4 n_2x2, 5 n_levels, 6 nnz(L), 7 U doubles per instance,
 *  8 factor multiply-adds per instance, 9 Schur multiply-adds per instance, 10 factor tasks,
 *  11 update runs, 12 Schur tiles, 13 Schur tile records, 14 canonical entries, 15 raw entries */
int pp_group_stats(pp_handle h, int group, int64_t out[16]);
/* More of the same.  out[16]: 0 raw entries read (rows of the transposed input), 1 doubles of packed pivot-block
 * inverses per instance, 2 doubles of pivot-block term magnitudes, 3 factor entries in coupling rows (what the Schur
 * kernel reads, per U and per L), 4 bytes of index data shared by all instances of the group, 5 forward-solve entries,
 * 6 coupling-row entries of the forward sweep, 7 rows of the source buffer (f2), 8 / 9 / 10 kernel launches of the
 * factorisation levels / forward sweep / backward sweep, 11 padded batch, 12 chunks of 64 instances, 13 Schur tiles,
 * 14 first level of the tail, 15 reserved. */
int pp_group_stats_ex(pp_handle h, int group, int64_t out[16]);
/* Elimination order of a group (new -> old), n ints. */
int pp_group_perm(pp_handle h, int group, int32_t* perm);
/* Inertia-correction fast path (interior_point.py:364-392, interfaces/interface.py:590-619,
 * sc_ip_interface.py:1736-1757): the regularised KKT differs from the one whose values are resident only by
 * + delta_w on the diagonal of the Hessian rows and - delta_c on the diagonal of the constraint rows.
 * pp_set_diagonal_classes (after pp_end_symbolic): cls[n] per row of K_i of the group -- 0 none, 1 Hessian row,
 * 2 constraint row; every classed row needs its diagonal entry in the planned pattern (status 3 otherwise).
 * pp_numeric_local_shifted = pp_numeric_local on the resident values with those shifts applied on the device
 * (no host staging, no H2D); the shift of the coupling block goes into Q of pp_factor_schur as usual. */
int pp_set_diagonal_classes(pp_handle h, int group, const int8_t* cls);
int pp_numeric_local_shifted(pp_handle h, double delta_w, double delta_c);
/* Host-side helper of the LinearSolverInterface boundary (no device work, no handle): stages the raw COO values of
 * `nblocks` blocks of one pattern group.  A block whose index arrays (kr, kc: K_i; br, bc: A_i; int32) equal the
 * group's reference arrays -- the common case: same entry order as at symbolic time -- has its values copied to row
 * slots[i] of `staging` (row_stride doubles per row: K values, then border values) and same_out[i] = 1; otherwise
 * same_out[i] = 0 and the caller canonicalises that block itself (the reference tolerates any entry order and
 * duplicates, quirk Q7: it calls .tocsr()/.tocoo() per block, mpi_...:294, 313-333).  Compare and copy are
 * memory-bound (0.9 GB per call at C3) and run on `nthreads` host threads. */
int pp_stage_values(int nblocks, int nthreads, const int32_t* const* kr, const int32_t* const* kc,
                    const double* const* kd, const int64_t* knnz, const int32_t* const* br, const int32_t* const* bc,
                    const double* const* bd, const int64_t* bnnz, const int32_t* ref_kr, const int32_t* ref_kc,
                    int64_t ref_knnz, const int32_t* ref_br, const int32_t* ref_bc, int64_t ref_bnnz, double* staging,
                    int64_t row_stride, const int32_t* slots, uint8_t* same_out);

/* The same restricted to runs of entries: runs_k / runs_b hold triples {first entry, length, destination offset in the
 * staging row} over the K data and the border data of a block -- the entries pp_used_raw_entries lists, so that the
 * staging rows are the compact rows pp_upload_values_compact takes. */
int pp_stage_values_runs(int nblocks, int nthreads, const int32_t* const* kr, const int32_t* const* kc,
                         const double* const* kd, const int64_t* knnz, const int32_t* const* br, const int32_t* const* bc,
                         const double* const* bd, const int64_t* bnnz, const int32_t* ref_kr, const int32_t* ref_kc,
                         int64_t ref_knnz, const int32_t* ref_br, const int32_t* ref_bc, int64_t ref_bnnz, int nruns_k,
                         const int64_t* runs_k, int nruns_b, const int64_t* runs_b, double* staging, int64_t row_stride,
                         const int32_t* slots, uint8_t* same_out);
/* pp_stage_values_runs + pp_upload_values_compact with the two overlapped: blocks (ascending slots; staging row =
 * slot, n_used doubles per row, pinned) are staged in slices on host threads and every finished slice is sent with an
 * asynchronous copy on the handle's stream while the next one is staged. */
int pp_stage_upload_compact(pp_handle h, int group, int nblocks, int nthreads, const int32_t* const* kr,
                            const int32_t* const* kc, const double* const* kd, const int64_t* knnz,
                            const int32_t* const* br, const int32_t* const* bc, const double* const* bd,
                            const int64_t* bnnz, const int32_t* ref_kr, const int32_t* ref_kc, int64_t ref_knnz,
                            const int32_t* ref_br, const int32_t* ref_bc, int64_t ref_bnnz, int nruns_k, const int64_t* runs_k,
                            int nruns_b, const int64_t* runs_b, double* staging, const int32_t* slots, uint8_t* same_out);
/* The same for blocks whose index arrays the caller has verified against the reference order at an earlier call (nothing
 * is compared; kd[i] / bd[i]: ref_knnz / ref_bnnz values), WITHOUT waiting: the arguments are copied, the host threads
 * stage the rows and send every slice they finish themselves, and the call returns, so that the caller prepares its next
 * batch of blocks meanwhile.  A second begin first waits for the job in flight; pp_stage_upload_end waits for the last one
 * and reports the first error.  No other call on the handle in between. */
int pp_stage_upload_verified_begin(pp_handle h, int group, int nblocks, int nthreads, const double* const* kd,
                                   const double* const* bd, int64_t ref_knnz, int64_t ref_bnnz, int nruns_k, const int64_t* runs_k,
                                   int nruns_b, const int64_t* runs_b, double* staging, const int32_t* slots);
int pp_stage_upload_end(pp_handle h);
/* dst[idx[i]][0 .. row_doubles) = src[i][0 .. row_doubles) on host threads: the right-hand sides of the local blocks into
 * the rows of their staging array (handle-free, no device work). */
int pp_copy_rows(int nrows, int nthreads, const double* const* src, const int64_t* idx, double* dst, int64_t row_doubles);
/* The right-hand sides of ALL blocks of a group (src[i]: the n values of the block in slot i, nrows = batch) through the
 * pinned staging array [batch][n] to the device; the copy of a slice of rows overlaps the host threads' work on the next
 * slice.  Replaces pp_copy_rows + pp_upload_rhs on the host boundary of do_back_solve (mpi_...:363-380 reads the blocks
 * of the right-hand side one by one). */
int pp_upload_rhs_rows(pp_handle h, int group, int nrows, int nthreads, const double* const* src, double* staging);
/* The solutions of a group to the host.  dst NULL: one asynchronous copy into the pinned array [batch][n] (synchronise,
 * pp_synchronize, before reading it).  dst given (pageable [batch][n], e.g. a fresh array per call as mpi_...:390-401
 * returns one): slices arrive in the pinned array and host threads move each on to dst while the next is in flight;
 * returns when dst is complete. */
int pp_download_solution_rows(pp_handle h, int group, int nthreads, double* pinned, double* dst);
/* Pinned (page-locked) host memory for the staging arrays and result buffers of the host boundary. */
void* pp_host_alloc(int64_t bytes);
void pp_host_free(void* p);

/* After a numeric factorisation that reported numerically zero pivots: the first instance (slot in the group's
 * batch) whose block broke down, earliest pivot in elimination order first, or -1 if no block of this group did.
 * The pivot sequence is static per pattern group and fixed from representative values at symbolic time; MA27, the
 * reference's sub-solver, pivots dynamically (ma27_interface.py:124-136 only reports singular for a matrix that
 * is), so the host class uses this to refresh the pivot order from the values that broke it and to factorise once
 * more before it reports `singular` to the inertia-correction loop. */
int pp_find_zero_pivot(pp_handle h, int group, int32_t* instance_out);

/* ---- f4: vector kernels of the step after the solve, on device-resident vectors ------------------
 * (parapint/algorithms/interior_point.py:174-317 check_convergence, :655-758 fraction_to_the_boundary, :619-626 the
 * step).  All pointers are device arrays of n doubles on the handle's device; NULL = that array is absent.
 * pp_vec_step_stats: one pass over a variable family (x with step dx, bounds xl / xu, bound duals zl / zu with steps
 * dzl / dzu) -> out = {alpha_primal, alpha_dual, max |(x - xl) zl - mu|, max |(xu - x) zu - mu|}: the
 * fraction-to-the-boundary step lengths min(1, min -tau (x - xl)/dx over dx < 0, min tau (xu - x)/dx over dx > 0) and
 * min(1, min -tau z/dz over dz < 0), and the complementarity residuals over the finite bounds.
 * pp_vec_max_abs: max |v_i| (the infeasibility norms).  pp_vec_axpy: y += alpha x (the step).  Stream-ordered after the
 * back-solve; the two reductions synchronise to hand their scalars to the host. */
int pp_vec_step_stats(pp_handle h, int64_t n, const double* x, const double* dx, const double* xl, const double* xu,
                      const double* zl, const double* dzl, const double* zu, const double* dzu, double tau, double mu,
                      double out_host[4]);
int pp_vec_max_abs(pp_handle h, int64_t n, const double* v, double* out_host);
int pp_vec_axpy(pp_handle h, int64_t n, double alpha, const double* x, double* y);
/* scatter != 0: dst[0..ndst) = 0, dst[idx[i]] = src[i]; else dst[i] = src[idx[i]] (i < n; idx: int64 on the device, the caller
 * guarantees its range).  The coupling block of a time-staged problem between the caller's ordering and the padded ordering
 * under which S is block tridiagonal (mpi_explicit_schur_complement.py:88-125). */
int pp_vec_permute(pp_handle h, int64_t n, const int64_t* idx, const double* src, double* dst, int64_t ndst, int scatter);

/* ---- f2 / f4: the interior-point step on device-resident iterates ---------------------------------------
 * What parapint's interfaces and ip_solve do around the linear solve in every iteration -- barrier diagonals of the KKT
 * matrix (interfaces/interface.py:450-465), right-hand side (:496-538, schur_complement/sc_ip_interface.py:1683-1696),
 * bound-dual steps (:562-588), fraction to the boundary (algorithms/interior_point.py:655-758), the step (:619-626) and
 * the three convergence measures (:174-317) -- for a two-stage stochastic QP whose scenarios sit in HBM in the solver's
 * own [row][instance] layout.  One pp_ip_group per pattern group of the solver (instance b = block slots[b]):
 *   W       iterate, rows x(n) | s(mi) | y_eq(me) | y_ineq(mi) | y_link(nfs) | z_l(n) | z_u(n) | s_l(mi) | s_u(mi): the first
 *           nb = n + 2 mi + me + nfs rows are in the row order of the KKT block (and of rhs / delta)
 *   bounds  lb(n) | ub(n) | ineq_lb(mi) | ineq_ub(mi) (+-inf = none; relaxed as interface.py:389-419)
 *   data    c(n) | b_eq(me)
 *   src     the source tensor the factorisation reads its values from (pp_bind_source_buffer): rows src_dp .. + n and
 *           src_ds .. + mi receive the barrier diagonals, the Hessian / Jacobian values are read from it
 *   G       [n][bpad] work rows: grad f + J^T y at the current iterate
 *   rhs     right-hand side [nb][bpad]; delta: the solution of the KKT system [nb][bpad] (pp_bind_native_vectors)
 *   prog    n + me + mi + nfs row programs {t0, tH, t1, o} for the rows grad_x L, A_eq x - b, A_ineq x - s, x_fs - z:
 *           terms [t0, t1) as pairs {source row or -1 (the constant 1), row of W}, the Hessian terms [t0, tH) first; o: the
 *           order of execution -- slot s works on row prog[4 s + 3] (a permutation; rows that share source entries close
 *           together keep the second read of an entry in the L2)
 * All arrays are [rows][bpad] doubles on the handle's device, bpad a multiple of 64, instances >= batch are padding
 * (they hold a copy of a real scenario without bounds and are left alone).  At most 8 groups.
 * Time-staged problems (sc_ip_interface.py:13-1026; the coupling block is [rho: multipliers of the forward links | z:
 * coupling states], ncz of each) use MAPPED groups, zoff != NULL ([2][bpad] int32): link row k of instance b ties coupling
 * state zoff[b] + k (the nfs rows inside the block: the backward link) resp. zoff[bpad + b] + k (nfw further rows: the
 * forward link, whose multipliers live in the coupling block -- W carries the instance's copy of them in nfw rows behind
 * the bound duals, prog has nfw more rows, their residuals go to the rho part of the coupling right-hand side).  Then z
 * has ncz entries, dz is the coupling solution [d rho | d z], and v_local / v_table rows are 8 + 2 ncz long:
 * {..., rho rows of the coupling right-hand side (this rank's forward links), z rows (this rank's link duals)}.  Without
 * a map every instance ties all nfs coupling variables (two-stage stochastic programs), nfw = 0 and ncz is ignored.
 * Nonlinear models (obj_row >= 0): the caller evaluates its model at the iterate before pp_ip_residuals and leaves grad f in
 * data rows 0 .. n - 1, -c_eq(x) in rows n .. n + me - 1, the objective value of every instance in row obj_row, the current
 * Hessian-of-the-Lagrangian and Jacobian values in src; the row programs then carry no Hessian terms.  obj_row = -1: a QP
 * (data = c | b_eq, objective 1/2 x'Hx + c'x from the row programs).
 *   pp_ip_rhs           rows x and s of rhs from G, the iterate and the barrier parameter mu (the other rows are written by
 *                       pp_ip_residuals)
 *   pp_ip_step_lengths  alpha_local[2] (device) = this rank's fraction-to-the-boundary step lengths, tau = 1 - mu; the
 *                       bound-dual steps are formed on the fly from delta
 *   pp_ip_take_step     iterate += alpha * step with alpha = min over the nranks rows of alpha_table (device, [nranks][2];
 *                       unified != 0: one length for both); alpha_table == NULL: no step (measures of the initial point).
 *                       z / dz: coupling variables and their step (device, nfs).  Writes the barrier diagonals into src.
 *   pp_ip_residuals     G, the constraint rows of rhs, and v_local (device, 8 + nfs): {primal infeasibility, dual
 *                       infeasibility of the primal rows, complementarity at 0 and at mu, sum |bound duals|, sum |constraint
 *                       duals|, objective, dual infeasibility of the slack rows, sum over the instances of y_link (nfs)}
 *   pp_ip_publish       combines the nranks rows of v_table ([nranks][8 + ncoup], device; rank order, deterministic) into the
 *                       coupling right-hand side rhs_coupling (device, ncoup = nfs, or 2 ncz for mapped groups) and the
 *                       mailbox; the entries from dual_from on (0, or ncz) are also -grad L of the coupling variables
 *   pp_ip_wait          blocks until the last pp_ip_publish has run: out = {primal inf, dual inf, compl(0), compl(mu),
 *                       sum |bound duals|, sum |duals|, objective, alpha_primal, alpha_dual, 0}
 * All calls are stream-ordered on the handle's stream; only pp_ip_wait synchronises (it polls a pinned mailbox).
 * Between ranks the caller all-gathers alpha_local -> alpha_table and v_local -> v_table (pp_comm_allgather or any other
 * transport); with one rank the tables are the local arrays. */
typedef struct pp_ip_group {
  int32_t n, mi, me, nfs, batch, bpad, src_dp, src_ds, nfw, ncz, obj_row, reserved;
  double* W;
  const double* bounds;
  const double* data;
  double* src;
  double* G;
  double* rhs;
  const double* delta;
  const int32_t* prog;
  const int32_t* terms;
  const int32_t* zoff;
} pp_ip_group;
int pp_ip_rhs(pp_handle h, int ngroups, const pp_ip_group* groups, double mu);
int pp_ip_step_lengths(pp_handle h, int ngroups, const pp_ip_group* groups, double tau, double mu, double* alpha_local);
int pp_ip_take_step(pp_handle h, int ngroups, const pp_ip_group* groups, const double* alpha_table, int nranks, int unified,
                    double mu, double* z, const double* dz);
int pp_ip_residuals(pp_handle h, int ngroups, const pp_ip_group* groups, const double* z, double* v_local);
int pp_ip_publish(pp_handle h, const double* v_table, const double* alpha_table, int nranks, int ncoup, int dual_from,
                  double* rhs_coupling);
int pp_ip_wait(pp_handle h, double out[10]);
/* With pp_profile(h, 1): accumulated device time (HIP events on the handle's stream), launches and calls of {pp_ip_rhs,
 * pp_ip_step_lengths, pp_ip_take_step, pp_ip_residuals}. */
int pp_ip_phase_times(pp_handle h, double ms_out[4], int32_t launches_out[4], int32_t calls_out[4]);

/* A caller's device model in HIP, shipped with the library for the example parapint_amd/examples/burgers.py (csrc/example_burgers.hip):
 * the functions of the discretised Burgers control problem (parapint/examples/burgers.py:63-176) for all time blocks of a pattern
 * group, on the arrays of pp_ip_group with obj_row >= 0 -- grad f and -c(x) into data, the objective of every lane into row
 * obj_row, Jacobian and Hessian-of-the-Lagrangian values into src.  stream: the HIP stream to enqueue on (the solver's);
 * dt_lane [bpad], w [nt + 1][bpad], y0 [m], scratch [(nt + 1) bpad] doubles on the device.  Not part of the solver. */
int pp_example_burgers_model(void* stream, int m, int nt, int n, int bpad, int y_eq, int hess, int jac, int obj_row,
                             int init_conditions, int start_term, double dx, double omega, double v, double r,
                             const double* dt_lane, const double* w, const double* y0, const double* W, double* src, double* data,
                             double* scratch);

/* Pivot tolerances (MA27 cntl(1), ma27_interface.py:36-47; examples/stochastic.py:120-124 uses 1e-6).
 *   u_symbolic  threshold of the static pivot choice at symbolic time: a 1x1 pivot is taken only if
 *               |d| >= u * max|row| on the representative values, else a 2x2 pivot (0: keep 0.01)
 *   u_runtime   growth guard of every numeric factorisation: MA27's test |d| >= u max|column| is equivalent to
 *               |l_ij| <= 1/u for the factor entries it produces, which every instance checks as it scales its rows;
 *               an instance with a larger entry is flagged, counted (pp_get_growth_count, summed over the ranks by the
 *               S all-reduce) and the factorisation reports status 2, so that the caller can refresh the pivot order
 *               from that instance (pp_find_growth) exactly as for a zero pivot, or regularise.  0 (the default): the
 *               guard is not enforced -- the benign 1/mu growth of interior-point matrices (a slack pivot of 1e-9
 *               against a -1 coupling) would otherwise be reported --, instances beyond 1e8 are only counted
 * Affects groups added / factorisations started afterwards. */
int pp_set_pivot_tolerance(pp_handle h, double u_symbolic, double u_runtime);
int pp_get_growth_count(pp_handle h, int64_t* out);
int pp_find_growth(pp_handle h, int group, int32_t* instance_out);

/* ---- collectives without a host hop ------------------------------------------------------
 * The two data-sized exchanges of the path -- the sum all-reduce of [S | status tail] once per numeric factorisation
 * (mpi_explicit_schur_complement.py:343, with :21 and :427-429 folded into the tail) and of r_s once per back-solve (:387)
 * -- enqueued by the library itself as RCCL calls on the handle's stream, between its own kernels.  librccl is opened at
 * run time (the library has no link-time dependency on it).
 *   pp_comm_unique_id   rank 0: 128 bytes for ncclCommInitRank; the caller broadcasts them (any side channel)
 *   pp_comm_init        collective over the nranks processes (one per GPU); replaces an earlier communicator
 *   pp_comm_size        ranks of the handle's communicator, 0 if none
 *   pp_allreduce_schur  after pp_numeric_local, before pp_factor_schur;  pp_allreduce_rs  after pp_solve_forward
 * Without a communicator the two calls return 3. */
int pp_comm_unique_id(uint8_t id_out[128]);
int pp_comm_init(pp_handle h, int nranks, int rank, const uint8_t id[128]);
int pp_comm_size(pp_handle h);
int pp_allreduce_schur(pp_handle h);
int pp_allreduce_rs(pp_handle h);
/* table[r * count .. ] = src of rank r (device arrays of doubles; the scalars of the interior-point step, above) */
int pp_comm_allgather(pp_handle h, const double* src, double* table, int64_t count);

/* Diagnostic: the factor of one instance (block) of a group after pp_numeric_local, in the plan's
 * panel storage: which = 0 unscaled panels U, 1 scaled rows L (the MA27 factor entries,
 * ma27_interface.py:124), 2 packed inverses of the block pivots.  count doubles are copied. */
int pp_get_factor(pp_handle h, int group, int which, int instance, double* out, int64_t count);

#ifdef __cplusplus
}
#endif
#endif /* PARAPINT_HIP_H */
