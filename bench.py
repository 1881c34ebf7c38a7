#!/usr/bin/env python
"""Benchmark of the hot path: interior-point iterations / second on the synthetic stochastic KKT.

One step = one IP-iteration-equivalent of linear algebra (reference call sites
parapint/algorithms/interior_point.py:553-567): ONE ``do_numeric_factorization`` on fresh values (same pattern) + ONE
``do_back_solve``, called through the ``LinearSolverInterface`` methods with the reference's keywords, including the
status / inertia read-back the IP loop needs.  Inputs are resident in HBM when the timed region starts: the matrix
is a ``DeviceBlockMatrix`` (the interface's own Hessian / Jacobian arrays as device sources + the value map fixed at
symbolic time), the right-hand side and the solution are ``DeviceBlockVector``s.  Reported beside it:

  boundary_host   the same two calls with host SciPy COO blocks in and host vectors out (staging + PCIe inside)
  device_only     the kernels driven through the C ABI without the Python class
  cpu_baseline    the reference algorithm (SuperLU sub-solver) on the host cores

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3|C2]

For N > 1 the script starts its own N ranks (``python -m torch.distributed.run``) unless it already runs under one.
Workload C3 = BASELINE.json configs[2] (the one the metric is quoted on): 1024 scenarios x (n_q=1000, n_y=4000 -> 5000
primal vars, block dim 9200), 200 coupling variables, sharded round-robin over the ranks (strong scaling).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = 'IP iterations/sec on 1024-scenario stochastic KKT (5k vars/block, 200 coupling)'
PHASES = ['assemble', 'factor_levels', 'schur_tiles', 'dense_S', 'fwd_levels', 'fwd_coupling', 'coupling_solve',
          'bwd_levels']
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_MFMA_PEAK_TF = 78.6    # MI355X public spec for fp64 matrix (= fp64 vector) throughput, SURVEY 8d
WORKLOADS = {'C3': (1024, 1000, 4, 200), 'C2': (64, 400, 4, 100),
             # BASELINE.json configs[3]: 512 time blocks x ~4k variables, n_s = 49 states, banded coupling (block-tridiagonal S)
             'C4': (512, 49, 2, 40),
             # BASELINE.json configs[4]: 4096 scenarios x 19 000 rows per KKT block, 1000 coupling variables (meant for 8 GPUs;
             # --blocks 512 is one rank's share)
             'C5': (4096, 2000, 4, 1000)}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='C3', choices=sorted(WORKLOADS))
    ap.add_argument('--blocks', type=int, default=0, help='override the number of scenario blocks')
    ap.add_argument('--scaling', default='strong', choices=['strong', 'weak'],
                    help='strong (default, BASELINE.json configs[2]: the blocks of the workload are sharded over the ranks) or '
                         'weak (every rank gets the full block count of the workload: N x world blocks in total)')
    ap.add_argument('--n-q', type=int, default=0)
    ap.add_argument('--m', type=int, default=0)
    ap.add_argument('--n-theta', type=int, default=0)
    ap.add_argument('--value-sets', type=int, default=6, help='distinct device-resident value sets cycled through')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-boundary', action='store_true')
    ap.add_argument('--no-ip-loop', action='store_true')
    ap.add_argument('--no-ip-loop-dynamic', action='store_true')
    ap.add_argument('--no-shares', action='store_true',
                    help='skip the one-rank shares of an 8-GPU run (C3: 128 blocks, C5: 512 blocks) the default line reports')
    ap.add_argument('--ip-loop-dynamic-all-ranks', action='store_true',
                    help='run the time-staged loop with more than one rank too (default: one rank only -- an auxiliary '
                         'measurement must not be able to stall the headline line of a multi-rank run)')
    ap.add_argument('--ip-time-blocks', type=int, default=512)
    ap.add_argument('--no-prefetch', action='store_true',
                    help='do not announce the right-hand side before the factorisation (solver.prefetch_forward)')
    ap.add_argument('--ip-scenarios', type=int, default=1024)
    ap.add_argument('--boundary-iterations', type=int, default=6)
    ap.add_argument('--result-buffers', type=int, default=2,
                    help='result_buffers of the solver: 2 = results handed out from two buffers in turn (what the timed loop and '
                         'the interior-point loops do), 0 = the class default, a fresh result per back-solve')
    ap.add_argument('--profile-steps', type=int, default=5)
    ap.add_argument('--sn-wmax', type=int, default=0, help='supernode width cap (0: library default)')
    ap.add_argument('--sn-tol', type=int, default=-1, help='padded rows tolerated when merging (-1: default)')
    return ap.parse_args()


def spawn_ranks(n):
    """Start n ranks of this script the way the driver would (one process per GPU, rendezvous on 127.0.0.1) and hand
    their exit code on.  Runs before anything touches the GPU."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    return subprocess.call(cmd, env=env)


def survey_bytes_per_block(z_K, z_L, n_i, n_c, z_A):
    """Algorithmic HBM bytes per block and iteration, SURVEY.md section 8(d) (tight Schur model B_sc')."""
    b_fac = 12 * z_K + 12 * z_L
    b_sc = 2 * 12 * z_L + 8 * n_c * n_c
    b_bs = 2 * (2 * 12 * z_L + 3 * 8 * n_i) + 2 * 12 * z_A
    return {'factor': b_fac, 'schur': b_sc, 'back_solve': b_bs, 'total': b_fac + b_sc + b_bs}


def build_bytes_per_block(st, ex, n_c, batch, fused_sources=True):
    """What THIS implementation has to move per block and iteration if every operand crossed HBM exactly once: values
    only -- the index data (task records, entry lists) is shared by all instances of a pattern group and amortised
    over the batch.  U and L panels are both stored (2 z_L); the Schur kernel reads only the coupling rows.  With the
    producer's source arrays read by the leaf kernels themselves (the default for a DeviceBlockMatrix) there is no
    assembly pass: the factorisation reads the nsrc source values instead of an assembled copy."""
    z_L, n = st['u_doubles'], st['n']
    idx = ex['index_bytes'] / max(1, batch)
    b = {
        'assemble': 0.0 if fused_sources else 8.0 * (ex['nsrc'] + ex['raw_used']),
        'factor_levels': 8.0 * ((ex['nsrc'] if fused_sources else ex['raw_used']) + 2 * z_L + 2 * z_L
                                + ex['dinv_doubles'] + 2 * ex['tm_doubles']) + idx,
        'schur_tiles': 8.0 * (2 * ex['coupling_entries']) + 8.0 * n_c * n_c * ex['nchunk'] / max(1, batch),
        'fwd_levels': 8.0 * (2 * n + ex['fwd_entries'] + 2 * n),
        'fwd_coupling': 8.0 * (ex['crow_entries'] + n),
        'bwd_levels': 8.0 * ((z_L - ex['tm_doubles']) + ex['dinv_doubles'] + n + 2 * n + 2 * n),
    }
    b['total'] = sum(b.values())
    return b


def host_boundary_section(solver, model, comm, world, dist, iterations, residual_check, expected_inertia):
    """SURVEY 8(d) to the letter: host SciPy COO blocks in, host vectors out -- and the two opt-in variants of handing the
    values over (constant entries declared; flat value vectors).  Returns (boundary, declared, flat_plain, residual, ok,
    seconds of the symbolic phase)."""
    boundary = declared = flat_plain = flat_declared = None
    ok = True
    kkt = model.build_kkt(comm=comm, iteration=0)
    rhs = model.build_rhs(comm=comm)
    t0 = time.perf_counter()
    solver.do_symbolic_factorization(kkt)
    t_symbolic_host = time.perf_counter() - t0
    solver.do_numeric_factorization(kkt)
    solver.do_back_solve(rhs)
    class _Phases(object):            # (the solver's timer labels, mpi_...:207-255 / 291-360 plus the boundary's own)
        def __init__(self):
            self.t, self.open = {}, {}

        def start(self, name):
            self.open[name] = time.perf_counter()

        def stop(self, name):
            self.t.setdefault(name, []).append(time.perf_counter() - self.open.pop(name))

    def boundary_loop(first, flat=False):
        ts = []
        phases = _Phases()
        kkt_it = x = None
        bufs = []
        prev = None
        for it in range(first, first + iterations):
            # (the matrix of the previous iteration stays referenced HERE across the timed calls: the solver drops its own
            # reference to it inside do_numeric_factorization, and releasing 1024 blocks x 290 KB -- one munmap each -- is
            # the producer's cost, not the solver's: 7 ms per iteration when it fell into the timed region)
            prev = kkt_it
            kkt_it = model.build_kkt(comm=comm, iteration=it)
            if flat:
                # (a producer keeps its value arrays: two of them filled in turn, OUTSIDE the timed calls -- a fresh 300 MB
                # array per iteration made the solver's release of the previous matrix, an munmap of that size, part of the
                # timed factorisation: 18 ms per iteration against 7.6 ms between the solver's own labels)
                fresh = model.flat_values(iteration=it)
                if len(bufs) < 2:
                    bufs.append(fresh)
                else:
                    bufs[it % 2][...] = fresh
                    fresh = bufs[it % 2]
                handed = HostValueMatrix(kkt, fresh)
            else:
                handed = kkt_it
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            solver.do_numeric_factorization(matrix=handed, raise_on_error=False, timer=phases)
            x = solver.do_back_solve(rhs, timer=phases)
            ts.append(time.perf_counter() - t0)
            prev = None
        med = float(np.median(ts))
        if world > 1:
            med = float(comm.allreduce_max(np.array([med]))[0])
        return med, ts, phases, kkt_it, x

    def phase_table(phases):
        return {k: round(1e3 * float(np.median(v)), 3) for k, v in phases.t.items()
                if k in ('values to device', 'factorize', 'form SC', 'factor SC', 'rhs to device', 'solve',
                         'solution to host', 'back_solve')}

    # the same loop with the entries that do not depend on the iteration declared constant by the producer (an
    # interface with linear constraints knows them: solver.declare_constant_entries) -- a second number, `value_boundary`
    # stays the undeclared one
    if hasattr(model, 'constant_entries') and hasattr(solver, 'declare_constant_entries'):
        solver.declare_constant_entries(model.constant_entries())
        med_d, ts_d, phases_d, kkt_d, x_d = boundary_loop(101)
        resid_d = residual_check(kkt_d, x_d, rhs)
        ok = ok and resid_d <= 1e-8 and tuple(solver.get_inertia()) == expected_inertia
        declared = {'it_per_s': 1.0 / med_d, 'ms_per_iteration': 1e3 * med_d, 'iterations': len(ts_d),
                    'phases_ms': phase_table(phases_d), 'residual': resid_d,
                    'note': 'as boundary_host, after solver.declare_constant_entries(...): the Jacobian and identity '
                            'entries of the synthetic KKT system are not compared or copied again (the staging threads get the runs of the other entries only)'}
        # ... and with the values handed over as the rows of one flat array over the symbolic phase's pattern object
        # (HostValueMatrix: one staging call per pattern group instead of a walk over 2 x N SciPy objects)
        if hasattr(model, 'flat_values'):
            from parapint_amd.sparse.host_value_matrix import HostValueMatrix
            med_f, ts_f, phases_f, kkt_f, x_f = boundary_loop(201, flat=True)
            resid_f = residual_check(kkt_f, x_f, rhs)
            ok = ok and resid_f <= 1e-8 and tuple(solver.get_inertia()) == expected_inertia
            flat_declared = {'it_per_s': 1.0 / med_f, 'ms_per_iteration': 1e3 * med_f, 'phases_ms': phase_table(phases_f),
                             'residual': resid_f}
            del kkt_f, x_f
        solver.declare_constant_entries(None)
        del kkt_d, x_d
    if hasattr(model, 'flat_values') and hasattr(model, 'constant_entries'):
        from parapint_amd.sparse.host_value_matrix import HostValueMatrix
        med_f, ts_f, phases_f, kkt_f, x_f = boundary_loop(301, flat=True)
        resid_f = residual_check(kkt_f, x_f, rhs)
        ok = ok and resid_f <= 1e-8 and tuple(solver.get_inertia()) == expected_inertia
        flat_plain = {'it_per_s': 1.0 / med_f, 'ms_per_iteration': 1e3 * med_f, 'iterations': len(ts_f),
                      'phases_ms': phase_table(phases_f), 'residual': resid_f,
                      'constant_declared': flat_declared,
                      'note': 'as boundary_host, the values handed over as HostValueMatrix(pattern, [blocks][entries]) '
                              '(opt-in container: flat value vectors over the pattern object of the symbolic phase); '
                              'constant_declared: the same after solver.declare_constant_entries(...)'}
        del kkt_f, x_f
    med, ts, phases, kkt_it, x = boundary_loop(1)
    boundary = {'it_per_s': 1.0 / med, 'ms_per_iteration': 1e3 * med, 'iterations': len(ts),
                'phases_ms': phase_table(phases),
                'note': 'host COO blocks in, host vectors out: needed entries staged into pinned memory on host '
                        'threads with the H2D overlapped, pinned D2H of x; median, max over ranks; phases_ms: host '
                        'wall time between the labels (rank 0; "factorize" contains "values to device", '
                        '"form SC" contains "factorize", "back_solve" its three parts)'}
    resid_boundary = residual_check(kkt_it, x, rhs)
    ok = ok and resid_boundary <= 1e-8 and tuple(solver.get_inertia()) == expected_inertia
    del kkt, kkt_it, x
    return boundary, declared, flat_plain, resid_boundary, ok, t_symbolic_host


def main():
    args = parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dynamic = args.workload == 'C4'
    N, n_q, m, n_t = WORKLOADS[args.workload]
    N, n_q, m, n_t = args.blocks or N, args.n_q or n_q, args.m or m, args.n_theta or n_t
    if args.scaling == 'weak':
        N *= world

    # ---- CPU baseline first: it forks worker processes, so it runs before the GPU is touched
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not dynamic:
        from oracle import cpu_baseline as cb
        cpu_baseline = cb.run(N, n_q, m, n_t, blocks_per_worker=max(1, min(64, N // 16)))   # ~10 s on 16 cores at C3

    import torch
    import torch.distributed as dist
    from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT, distribute_blocks
    from parapint_amd.linalg.comm import SerialComm, TorchComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
    from parapint_amd.linalg.results import LinearSolverStatus

    # Rehearsal switch (not the measured configuration): PP_BENCH_REHEARSAL=gloo lets several ranks share the GPUs
    # that are present and exchange through gloo (host-staged all-reduce), to exercise the N > 1 code path on a
    # one-GPU box.  The driver's multi-GPU runs use one GPU per rank and RCCL.
    rehearsal = os.environ.get('PP_BENCH_REHEARSAL', '')
    if rehearsal:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    elif world > torch.cuda.device_count():
        # (one process per GPU over RCCL: two ranks on one device cannot form a communicator -- say so instead of failing
        # inside ncclCommInitRank; PP_BENCH_REHEARSAL=gloo shares the devices that are there over gloo)
        sys.exit('bench.py --gpus %d: only %d HIP device(s) visible (one rank per GPU; PP_BENCH_REHEARSAL=gloo rehearses '
                 'several ranks on fewer devices over gloo)' % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    backend = 'none'
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if rehearsal:
            dist.init_process_group(rehearsal, rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        comm = TorchComm()
        backend = comm.backend
    else:
        comm = SerialComm()
    dev = torch.device('cuda', local_rank)

    local = distribute_blocks(N, rank, world)
    if dynamic:
        from parapint_amd.examples.performance.schur_complement.dynamic_kkt import SyntheticDynamicKKT
        model = SyntheticDynamicKKT(N, n_q, m, n_t, local_blocks=local)          # (T, n_s, n_u, nfe)
        n_c = model.n_coupling
        expected_inertia = (N * model.n_x + n_q * (N - 1), N * model.n_eq + 2 * n_q * (N - 1), 0)
        host_kkt = lambda it: model.build_kkt(comm=comm, iteration=it)       # noqa: E731
        set_sources = lambda s: {ndx: model.block_sources(ndx, 100 + s) for ndx in local}     # noqa: E731
        host_kkt_of_set = lambda s, srcs: model.build_kkt(comm=comm, iteration=100 + s)       # noqa: E731
        describe = ('%s: %d time blocks x %d variables (n_s=%d states, n_u=%d controls, %d elements; block dim %d), '
                    '%d coupling variables (forward-link duals + coupling states), block-tridiagonal S' %
                    (args.workload, N, model.n_x, n_q, m, n_t, model.block_dim(1), n_c))
    else:
        model = SyntheticKKT(N, n_q, m, n_t, local_blocks=local)
        n_c = n_t
        expected_inertia = (N * (model.n_y + n_q) + n_t, N * (model.n_y + n_t), 0)
        host_kkt = lambda it: model.build_kkt(comm=comm, iteration=it)       # noqa: E731
        w_entry = np.linspace(0.5, 1.5, model.n_y)

        def set_sources(s):       # fresh Hessian values for every block, every entry and every set
            out = {}
            for ndx in local:
                src = model.block_sources(ndx, None)
                eps = np.random.default_rng(10_000 * (100 + s) + ndx).uniform(0.0, 0.5)
                src[:model.n_y] = 2.0 + eps * w_entry
                out[ndx] = src
            return out
        host_kkt_of_set = lambda s, srcs: model.build_kkt_from_sources(srcs, comm=comm)       # noqa: E731
        describe = ('%s: %d scenario blocks x (n_q=%d, n_y=%d: %d primal vars, block dim %d), %d coupling vars' %
                    (args.workload, N, n_q, model.n_y, n_q + model.n_y, model.block_dim, n_t))
    B = len(local)
    solver = HipSchurComplementLinearSolver({i: None for i in local}, None, comm=comm, result_buffers=args.result_buffers)
    eng = solver._eng
    lib, h = eng.lib, eng.ns.h
    if args.sn_wmax > 0 or args.sn_tol >= 0:
        eng.set_supernodes(args.sn_wmax, args.sn_tol)

    def flat(v):
        return v.flatten() if hasattr(v, 'get_block') else np.asarray(v, dtype=np.double).ravel()

    def residual_check(kkt, x, rhs):
        """max over the local block rows and the coupling rows of |Kx - b| / (|K|_inf |x|_inf + |b|_inf)."""
        worst = 0.0
        xc = flat(x.get_block(N))
        rc_local = np.zeros(n_c)
        for ndx in local:
            K = kkt.get_block(ndx, ndx).tocsr()
            A = kkt.get_block(N, ndx).tocsr()
            xi, r = flat(x.get_block(ndx)), flat(rhs.get_block(ndx))
            res = K @ xi + A.T @ xc - r
            scale = abs(K).sum(axis=1).max() * max(np.abs(xi).max(), np.abs(xc).max()) + np.abs(r).max()
            worst = max(worst, float(np.abs(res).max() / scale))
            rc_local += A @ xi
        rc = comm.allreduce_sum(rc_local) if world > 1 else rc_local
        Qb = kkt.get_block(N, N)
        rc = rc + (Qb.tocsr() @ xc if Qb is not None else 0.0) - flat(rhs.get_block(N))
        worst = max(worst, float(np.abs(rc).max() / (np.abs(xc).max() * N + 1.0)))
        if world > 1:
            worst = float(comm.allreduce_max(np.array([worst]))[0])
        return worst

    # ---- (1) host boundary: SciPy COO blocks in, host vectors out
    boundary = None
    declared = None
    flat_plain = None
    resid_boundary = None
    ok = True
    if not args.no_boundary:
        boundary, declared, flat_plain, resid_boundary, ok, t_symbolic_host = host_boundary_section(
            solver, model, comm, world, dist, args.boundary_iterations, residual_check, expected_inertia)

    # ---- (2) the measured path: device-resident matrix and vectors through the LinearSolverInterface methods
    dkkt = model.build_device_kkt(comm=comm)
    t0 = time.perf_counter()
    res = solver.do_symbolic_factorization(matrix=dkkt, raise_on_error=False)
    t_symbolic = time.perf_counter() - t0
    assert res.status == LinearSolverStatus.successful
    gmain = max(range(len(solver.plan_stats)), key=lambda i: solver.plan_stats[i]['batch'])     # the largest group
    st = solver.plan_stats[gmain]
    ex = eng.ns.group_stats_ex(gmain)
    nsets = max(2, min(args.value_sets, args.steps + args.warmup))
    sets = []
    for sidx in range(nsets):
        srcs = set_sources(sidx)
        tensors = {}
        for gid, blocks in dkkt.slots.items():
            host = np.zeros(tuple(dkkt.sources[gid].shape))
            for b, ndx in enumerate(blocks):
                host[:, b] = srcs[ndx]
            tensors[gid] = torch.from_numpy(host).to(dev)
        sets.append((dkkt.with_sources(tensors), srcs if sidx == (args.warmup + args.steps - 1) % nsets else None))
    rhs_host = model.build_rhs(comm=comm)
    rhs_dev = solver.device_vector_from_host(rhs_host)

    def step(k, prefetch=None):
        # (as in an interior-point iteration, interior_point.py:553-566, the right-hand side exists before the matrix is
        # factorised: announcing it lets its forward sweep run beside the dense factorisation of S.  That call is an
        # extension of this package's solver class -- ip_solve / ip_solve_device of this package make it behind hasattr --;
        # the same loop in the reference's plain call order is timed below and reported as `value_no_prefetch`)
        if (not args.no_prefetch) if prefetch is None else prefetch:
            solver.prefetch_forward(rhs_dev)
        r = solver.do_numeric_factorization(matrix=sets[k % nsets][0], raise_on_error=False)
        xd = solver.do_back_solve(rhs_dev)
        return r, xd

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for k in range(args.warmup):
        step(k)
    sync_all()
    stamps = np.zeros(args.steps + 1)
    stamps[0] = t0 = time.perf_counter()
    for k in range(args.steps):
        res, xd = step(args.warmup + k)
        stamps[k + 1] = time.perf_counter()
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        elapsed = float(comm.allreduce_max(np.array([elapsed]))[0])
    ms_per_step = 1e3 * elapsed / args.steps
    value = args.steps / elapsed
    median_ms = 1e3 * float(np.median(np.diff(stamps)))
    # the same K steps in the reference's unchanged call order (do_numeric_factorization, then do_back_solve: no
    # announcement of the right-hand side) -- what a caller that is not aware of the extension sees
    if args.no_prefetch:
        value_no_prefetch, ms_no_prefetch = value, ms_per_step
    else:
        solver.prefetch_forward(None)
        for k in range(min(3, args.warmup)):
            step(k, prefetch=False)
        sync_all()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(args.warmup + k, prefetch=False)
        sync_all()
        el_np = time.perf_counter() - t0
        if world > 1:
            el_np = float(comm.allreduce_max(np.array([el_np]))[0])
        value_no_prefetch, ms_no_prefetch = args.steps / el_np, 1e3 * el_np / args.steps
        res, xd = step(args.warmup + args.steps - 1)          # (the checked step below is the announced form again)
        sync_all()
    # the same K announced steps WITHOUT the a-posteriori check of the back-solves (solver.residual_check = False: what
    # rounds 1-5 timed as `value`): the price of never handing out an unchecked solution is value_unchecked - value
    solver.residual_check = False
    for k in range(min(3, args.warmup)):
        step(k)
    sync_all()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    sync_all()
    el_un = time.perf_counter() - t0
    if world > 1:
        el_un = float(comm.allreduce_max(np.array([el_un]))[0])
    solver.residual_check = True
    res, xd = step(args.warmup + args.steps - 1)              # (the checked step below is a checked one again)
    sync_all()
    solution_check = {'on': True, 'value_unchecked': args.steps / el_un, 'ms_per_step_unchecked': 1e3 * el_un / args.steps,
                      'cost_ms_per_step': ms_per_step - 1e3 * el_un / args.steps,
                      'backward_error_last_step': solver.last_residual, 'refine_tolerance': solver.refine_tolerance,
                      'residual_tolerance': solver.residual_tolerance, 'solves_refined': solver.solves_refined,
                      'refinement_steps': solver.refinement_steps, 'solve_repairs': solver.solve_repairs,
                      'collectives_per_step': 3 if world > 1 else 0,        # [S | status], r_s, the agreement of this check
                      'check_collective': (None if world == 1 else 'library RCCL all-reduce on the solver stream'
                                           if (solver._btd is None and eng._direct_rccl(comm)) else 'host all-reduce through the communicator'),
                      'note': 'every do_back_solve ends with the residual of all block rows and of the coupling rows on the '
                              'device (csrc/refine.hip) and one host read; refinement / a new pivot sequence follow only above '
                              'the tolerances (parapint_amd/linalg/solution_check.py)'}
    mem_max, _, mem_now = eng.memory_info()        # value storage: what the plan needs at most / what this path allocated

    # correctness of the last timed step: download and check against the assembled system
    k_last = (args.warmup + args.steps - 1) % nsets
    xh = xd.to_host(rhs_host)
    resid = residual_check(host_kkt_of_set(k_last, sets[k_last][1]), xh, rhs_host)
    inertia = solver.get_inertia()
    ok = ok and resid <= 1e-8 and res.status == LinearSolverStatus.successful and tuple(inertia) == expected_inertia

    # ---- (3) device only: the same kernels driven through the C ABI without the Python class
    qcorner = solver._btd_corner(dkkt.Q) if solver._btd is not None else None
    qdense = None if (dkkt.Q is None or solver._btd is not None) else (dkkt.Q.toarray() if hasattr(dkkt.Q, 'toarray') else dkkt.Q)

    def raw_step(k):
        for gid, t in sets[k % nsets][0].sources.items():
            eng.bind_source_tensor(gid, t)
        eng.numeric_factor_blocks()
        eng.numeric_schur(side=(not args.no_prefetch) and (world == 1 or eng._direct_rccl(comm)))
        eng.allreduce_schur(comm)
        if solver._btd is not None:
            eng.factor_schur_corner(*qcorner)
        else:
            eng.factor_schur(qdense)
        if not args.no_prefetch:                  # (forward sweep beside the dense phase, status read behind it)
            eng.solve_forward()
            eng.allreduce_rs(comm)
        status = eng.status()
        if args.no_prefetch:
            eng.solve_forward()
            eng.allreduce_rs(comm)
        eng.solve_coupling_dev(None)
        eng.solve_backward()
        return status
    for k in range(3):
        raw_step(k)
    sync_all()
    t0 = time.perf_counter()
    for k in range(args.steps):
        raw_step(k)
    sync_all()
    el = time.perf_counter() - t0
    if world > 1:
        el = float(comm.allreduce_max(np.array([el]))[0])
    device_only = {'it_per_s': args.steps / el, 'ms_per_step': 1e3 * el / args.steps}

    # ---- per-phase device time (HIP events on the solver's stream), separate untimed pass
    import ctypes
    phases = {}
    collective_us = None
    if args.profile_steps > 0:
        eng.ns.check(lib.pp_profile(h, 1), 'pp_profile')
        for k in range(args.profile_steps):
            step(k)
        ms = np.zeros(8)
        launches = np.zeros(8, dtype=np.int32)
        calls = np.zeros(8, dtype=np.int32)
        eng.ns.check(lib.pp_phase_times(h, ms.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                        launches.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                        calls.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))), 'pp_phase_times')
        eng.ns.check(lib.pp_profile(h, 0), 'pp_profile')
        # the two data-path collectives by themselves (events on the stream they are enqueued on, around the call the
        # solver makes -- torch.distributed or the library's own RCCL call): this rank's average and every rank's
        if world > 1:
            marks = {'schur': [], 'rs': []}
            plain = {'schur': eng.allreduce_schur, 'rs': eng.allreduce_rs}

            def timed(name):
                def call(c):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    plain[name](c)
                    e1.record()
                    marks[name].append((e0, e1))
                return call
            eng.allreduce_schur, eng.allreduce_rs = timed('schur'), timed('rs')
            for k in range(max(3, args.profile_steps)):
                step(k)
            torch.cuda.synchronize(dev)
            eng.allreduce_schur, eng.allreduce_rs = plain['schur'], plain['rs']
            mine = torch.tensor([1e3 * float(np.mean([a.elapsed_time(b) for a, b in marks[k][1:]])) for k in ('schur', 'rs')],
                                dtype=torch.float64, device=dev)
            if backend == 'nccl':
                allr = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(allr, mine)
                per_rank = [[float(v) for v in t.cpu()] for t in allr]
            else:
                per_rank = [[float(v) for v in row] for row in comm.allgather(mine.cpu().numpy())]
            collective_us = {'allreduce_S_and_status': [r[0] for r in per_rank], 'allreduce_r_s': [r[1] for r in per_rank],
                             'bytes': [8 * (eng.schur_doubles + 8), 8 * solver._nc]}
        for i, name in enumerate(PHASES):
            if calls[i] > 0:
                phases[name] = {'ms_per_step': float(ms[i] / args.profile_steps),
                                'launches_per_step': int(launches[i] // args.profile_steps)}

    # ---- roofline of the dominant kernel class (by device time)
    z_K = st['canonical_entries']
    z_L = st['u_doubles']
    nc_blk = st['n_coupling']
    sb = survey_bytes_per_block(z_K, z_L, st['n'], nc_blk, nc_blk)
    fused = phases.get('assemble', {}).get('launches_per_step', 0) == 0     # no assembly kernel ran
    bb = build_bytes_per_block(st, ex, nc_blk, B, fused_sources=fused)
    survey_phase = {'assemble': bb['assemble'], 'factor_levels': float(sb['factor']),
                    'schur_tiles': float(sb['schur']), 'fwd_levels': sb['back_solve'] / 2.0,
                    'bwd_levels': sb['back_solve'] / 2.0}
    roofline = None
    if any(p in survey_phase for p in phases):
        dom = max((p for p in phases if p in survey_phase), key=lambda p: phases[p]['ms_per_step'])
        dom_ms = phases[dom]['ms_per_step']
        dom_launches = max(1, phases[dom]['launches_per_step'])
        bytes_per_launch = survey_phase[dom] * B / dom_launches
        achieved = bytes_per_launch / (dom_ms / dom_launches * 1e-3) / 1e9
        # HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of
        # this same command; profiles/pmc_traffic.json) -- valid for the build and workload it was collected on
        traffic, traffic_source = None, None
        try:
            from parapint_amd._native import kernel_source_sha1
            pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
            same_build = pmc.get('kernel_source_sha1') == kernel_source_sha1()
            traffic_source = {'file': 'profiles/pmc_traffic.json', 'kernel_source_sha1': pmc.get('kernel_source_sha1'),
                              'taken_on_this_build': same_build}
            ph = pmc['phases'].get(dom)
            if (same_build and ph and ph['launches_per_step'] > 0 and world == 1 and args.workload == 'C3'
                    and (N, n_q, m, n_t) == WORKLOADS['C3']):
                traffic = ph['hbm_bytes_per_step'] / ph['launches_per_step']
        except Exception:
            traffic = None
        per_phase = {}
        for p in phases:
            if bb.get(p, 0) > 0 and phases[p]['ms_per_step'] > 0:
                gbps = bb[p] * B / (phases[p]['ms_per_step'] * 1e-3) / 1e9
                per_phase[p] = {'GBps_build_model': gbps, 'frac': gbps / HBM_PEAK_GBS}
        roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_source,
                    'algorithmic_bytes_per_launch': bytes_per_launch, 'avg_launch_us': 1e3 * dom_ms / dom_launches,
                    'model': 'SURVEY.md 8(d): B_fac = 12 z_K + 12 z_L etc.; build_model = the bytes this implementation '
                             'moves if every operand crosses HBM once (index data shared by the batch)',
                    'frac_model': 'survey',            # (frac above: SURVEY 8(d) bytes; build_model below: this build's own count)
                    'build_model': {'bytes_per_launch': bb[dom] * B / dom_launches,
                                    'achieved': bb[dom] * B / (dom_ms * 1e-3) / 1e9,
                                    'frac': bb[dom] * B / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                    'phases_build_model': per_phase,
                    'whole_iteration': {'survey_bytes': sb['total'] * B, 'build_bytes': bb['total'] * B,
                                        'GBps_survey': sb['total'] * B / (ms_per_step * 1e-3) / 1e9,
                                        'GBps_build': bb['total'] * B / (ms_per_step * 1e-3) / 1e9,
                                        'frac_build_model': bb['total'] * B / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        'frac_survey_model': sb['total'] * B / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS}}
        wi = roofline['whole_iteration']
        if wi['frac_survey_model'] > 1.0:
            # (n_c = 1000: the survey model counts an 8 n_c^2 write of S per block that the batch-reduced Schur kernel never
            # performs -- a fraction above 1 is not evidence; only the build-model figure is reported then)
            wi['frac_survey_model'] = None
            wi['GBps_survey'] = None
            wi['note'] = 'survey model exceeds the HBM peak for this shape (it counts a per-block S write that is not performed): build model only'

    # dense phase (factorisation of S, replicated on every rank): fp64 MFMA work, SURVEY 8d F_S = n_c^3/3 + 4 n_c^2
    dense_phase = None
    if 'dense_S' in phases and n_c > 0 and solver._btd is None:
        f_s = n_c ** 3 / 3.0 + 4.0 * n_c ** 2
        tf = f_s / (phases['dense_S']['ms_per_step'] * 1e-3) / 1e12
        dense_phase = {'flops': f_s, 'ms': phases['dense_S']['ms_per_step'], 'achieved_TFLOPs': tf,
                       'peak_fp64_mfma_TFLOPs': FP64_MFMA_PEAK_TF, 'frac': tf / FP64_MFMA_PEAK_TF,
                       'note': 'one workgroup, latency-bound chain of n_c/16 panels; replicated on every rank'}

    # ---- interior-point loop with device-resident iterates: the C3-shaped stochastic QP (1024
    # scenarios x 5000 primal variables with bounds x 200 first-stage variables: KKT blocks of dimension 9200) through
    # ip_solve_device on every rank -- scenarios dealt round-robin, the KKT values, the right-hand side, the step and the
    # convergence measures never leave HBM and are produced by the library's own kernels; the inertia-correction retries
    # run from the resident values.  The literal metric: interior-point iterations per second.
    ip_loop = None
    if not args.no_ip_loop and args.workload == 'C3' and not args.blocks:
        from parapint_amd.algorithms.device_interior_point import ip_solve_device
        from parapint_amd.algorithms.interior_point import IPOptions, InteriorPointStatus
        from parapint_amd.examples.stochastic_qp import c3_stochastic_qp
        from parapint_amd.interfaces.schur_complement.device_sc_ip_interface import DeviceStochasticQPInterface
        nsc = args.ip_scenarios * (world if args.scaling == 'weak' else 1)
        mine = [i for i in range(nsc) if i % world == rank]
        qps, fsi = c3_stochastic_qp(nsc, n_q=n_q, m=m, n_theta=n_t, seed=1, local=mine)
        best = None
        for rep in range(2):                 # (the second run: the library's scratch and the allocator's pools exist)
            ipi = DeviceStochasticQPInterface(qps, fsi, comm=comm)
            ipo = IPOptions()
            ipo.linalg.solver = HipSchurComplementLinearSolver({i: None for i in mine}, None, comm=comm, result_buffers=2)
            hist, ipst = [], {}
            sync_all()
            t0 = time.perf_counter()
            ip_status, ip_iters = ip_solve_device(ipi, ipo, history=hist, stats=ipst)
            sync_all()
            t_ip = time.perf_counter() - t0
            loop_s = ipst['loop_s']
            if world > 1:
                tt = torch.tensor([loop_s, t_ip], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                loop_s, t_ip = float(tt[0]), float(tt[1])
            sv = ipo.linalg.solver
            cur = {'it_per_s': ip_iters / loop_s, 'iterations': ip_iters, 'ms_per_iteration': 1e3 * loop_s / max(ip_iters, 1),
                   'median_ms_per_iteration': 1e3 * float(np.median(ipst['iteration_s'])) if ipst.get('iteration_s') else None,
                   'loop_seconds': loop_s, 'setup_seconds': t_ip - loop_s, 'it_per_s_whole_call': ip_iters / t_ip,
                   'converged': ip_status == InteriorPointStatus.optimal,
                   'final_infeasibilities': list(hist[-1][:3]) if hist else None,
                   'scenarios': nsc, 'scenarios_per_gpu': len(mine), 'primal_variables_per_scenario': ipi.pattern_groups[0].n,
                   'block_dim': ipi.pattern_groups[0].nb, 'n_coupling': ipi.nfs,
                   'torch_ops_per_iteration': (ipst.get('torch_ops') or 0) / max(ip_iters, 1),
                   'inertia_retries_from_resident_values': sv.diagonal_shift_refactorizations,
                   'pivot_order_refreshes': sv.pivot_order_refreshes, 'refresh_causes': dict(sv.refresh_causes),
                   # collectives of one iteration: the all-reduce of [S | status] and of r_s (the solver's) + two all-gathers
                   # of a handful of scalars (step lengths; convergence measures + this rank's coupling right-hand side) --
                   # two dependent reductions with the step between them; rccl_ranks > 0: all four enqueued by the library
                   'collectives_per_iteration': 4 if world > 1 else 0,
                   'rccl_ranks': int(sv._eng.lib.pp_comm_size(sv._eng.ns.h)),
                   'note': 'it_per_s: iterations / wall time of the loop (barrier diagonals, right-hand side, numeric '
                           'factorisation with its inertia check, back-solve, step lengths, step, convergence measures), max '
                           'over ranks; the one-off symbolic phase and set-up are setup_seconds (it_per_s_whole_call includes them)'}
            if best is None or cur['it_per_s'] > best['it_per_s']:
                best = cur
            del ipi, ipo, sv
        ip_loop = best
        ok = ok and ip_loop['converged']
        # where an iteration goes (a third, untimed run with the library's phase events on: no dense / forward overlap while
        # they are) and the step kernels against the HBM roof: algorithmic bytes = every array a kernel has to read or write
        # once, x instances of this rank
        ipi = DeviceStochasticQPInterface(qps, fsi, comm=comm)
        ipo = IPOptions()
        ipo.linalg.solver = HipSchurComplementLinearSolver({i: None for i in mine}, None, comm=comm, result_buffers=2)
        slib, sh = ipo.linalg.solver._eng.lib, ipo.linalg.solver._eng.ns.h
        slib.pp_profile(sh, 1)
        _, prof_iters = ip_solve_device(ipi, ipo)
        sync_all()
        ms8, l8, c8 = np.zeros(8), np.zeros(8, dtype=np.int32), np.zeros(8, dtype=np.int32)
        ms4, l4, c4 = np.zeros(4), np.zeros(4, dtype=np.int32), np.zeros(4, dtype=np.int32)
        dp, ip32 = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)
        slib.pp_phase_times(sh, ms8.ctypes.data_as(dp), l8.ctypes.data_as(ip32), c8.ctypes.data_as(ip32))
        slib.pp_ip_phase_times(sh, ms4.ctypes.data_as(dp), l4.ctypes.data_as(ip32), c4.ctypes.data_as(ip32))
        slib.pp_profile(sh, 0)
        rows = {'rhs': 0.0, 'step_lengths': 0.0, 'take_step': 0.0, 'residuals': 0.0}
        for gs in ipi.states:
            pg = gs.pg
            nv, ny = pg.n + pg.mi, pg.me + pg.nfs
            rows['rhs'] += gs.B * 5.0 * nv
            rows['step_lengths'] += gs.B * 6.0 * nv
            rows['take_step'] += gs.B * (10.0 * nv + 3.0 * pg.mi + 3.0 * ny)
            rows['residuals'] += gs.B * (pg.nnzH + 2.0 * (pg.nnzAe + pg.nnzAi) + (pg.n + pg.me + 2 * pg.mi + pg.nfs)
                                         + (pg.n + pg.me) + 3.0 * pg.n + pg.n + (pg.me + pg.mi + pg.nfs))
        step_kernels = {}
        for i, name in enumerate(('rhs', 'step_lengths', 'take_step', 'residuals')):
            if c4[i] > 0 and ms4[i] > 0:
                t_ms = float(ms4[i]) / int(c4[i])
                gbps = 8.0 * rows[name] / (t_ms * 1e-3) / 1e9
                step_kernels[name] = {'ms': t_ms, 'algorithmic_MB': 8.0 * rows[name] / 1e6, 'GBps': gbps, 'frac_of_hbm_peak': gbps / HBM_PEAK_GBS}
        ip_loop['step_kernels'] = step_kernels
        ip_loop['step_kernels_ms_per_iteration'] = float(sum(v['ms'] for v in step_kernels.values()))
        ip_loop['solver_kernels_ms_per_iteration'] = float(ms8.sum()) / max(int(c8[1]), 1)
        ip_loop['profiled_run'] = {'iterations': prof_iters, 'numeric_factorizations': int(c8[1]), 'back_solves': int(c8[7])}
        del ipi, ipo, qps

    # ---- the same loop for a time-staged problem at the dimensions of BASELINE.json configs[3] (C4: 512 time blocks, 49
    # states between them, coupling block 2 * 49 * 511 = 50 078 with block-tridiagonal S): diffusion control, every time
    # block a QP of 2089 primal variables (KKT block 4254), iterates resident and rank-distributed by time block
    # (DeviceDynamicQPInterface; SURVEY.md section 8 rows f2 + f3)
    # (what the line reports from the main handle; the handle itself is released in front of the time-staged loops: they run
    # three pattern groups on three streams, and the streams of a live handle beside them share the hardware queues --
    # measured: the Burgers loop 10.3 ms per iteration with the handle alive, 8.3 ms without)
    rccl_ranks = int(lib.pp_comm_size(h))
    bcr_paths = (dict(zip(('unpivoted_ldl_on_matrix_cores', 'bunch_kaufman'), eng.bcr_block_paths()))
                 if solver._btd is not None else None)
    if not args.no_ip_loop_dynamic and not args.no_ip_loop:
        import gc
        sync_all()
        eng.ns.close()
        gc.collect()
    ip_loop_dynamic = None
    if not args.no_ip_loop_dynamic and not args.no_ip_loop and args.workload in ('C3', 'C4') and not args.blocks \
            and (world == 1 or args.ip_loop_dynamic_all_ranks):
        from parapint_amd.algorithms.device_interior_point import ip_solve_device
        from parapint_amd.algorithms.interior_point import IPOptions, InteriorPointStatus
        try:
            from parapint_amd.examples import dynamics_qp
            from parapint_amd.interfaces.schur_complement.device_sc_ip_interface import DeviceDynamicQPInterface
            Tb = args.ip_time_blocks * (world if args.scaling == 'weak' else 1)
            ns_d, nu_d, nfe_d = 49, 2, 40
            dargs = dict(nfe_per_block=nfe_d, n_states=ns_d, n_controls=nu_d, nu=0.15 / (ns_d + 1) ** 2 * Tb * nfe_d)
            mine_t = [t for t in range(Tb) if t % world == rank]
            tblocks = dynamics_qp.DiffusionControl.time_blocks(0.0, 1.0, Tb, local=mine_t, **dargs)
            best = None
            for rep in range(2):
                ipi = DeviceDynamicQPInterface(tblocks, comm=comm)
                ipo = IPOptions()
                ipo.linalg.solver = HipSchurComplementLinearSolver({t: None for t in mine_t}, None, comm=comm, result_buffers=2)
                hist, ipst = [], {}
                sync_all()
                t0 = time.perf_counter()
                ip_status, ip_iters = ip_solve_device(ipi, ipo, history=hist, stats=ipst)
                sync_all()
                t_ip = time.perf_counter() - t0
                loop_s = ipst['loop_s']
                if world > 1:
                    tt = torch.tensor([loop_s, t_ip], dtype=torch.float64, device=dev)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    loop_s, t_ip = float(tt[0]), float(tt[1])
                pg = max(ipi.pattern_groups, key=lambda g: len(g.members))
                cur = {'it_per_s': ip_iters / loop_s, 'iterations': ip_iters, 'ms_per_iteration': 1e3 * loop_s / max(ip_iters, 1),
                       'loop_seconds': loop_s, 'setup_seconds': t_ip - loop_s, 'converged': ip_status == InteriorPointStatus.optimal,
                       'final_infeasibilities': list(hist[-1][:3]) if hist else None, 'objective': ipi.evaluate_objective(),
                       'time_blocks': Tb, 'time_blocks_per_gpu': len(mine_t), 'states': ns_d, 'primal_variables_per_block': pg.n,
                       'block_dim': pg.nb, 'n_coupling': 2 * ipi.ncz,
                       'torch_ops_in_the_loop': ipst.get('torch_ops') or 0}
                if best is None or cur['it_per_s'] > best['it_per_s']:
                    best = cur
                del ipi, ipo
            ip_loop_dynamic = best
            del tblocks
        except Exception as exc:     # (an auxiliary measurement: reported, never a reason to lose the headline line)
            ip_loop_dynamic = {'converged': False, 'error': '%s: %s' % (type(exc).__name__, exc)}

    # ---- BASELINE.json configs[3] to the letter: the Burgers discretisation (parapint/examples/burgers.py) with 512 time blocks x
    # 4018 variables (nfe_x = 50, 40 time steps per block), 49 states between the blocks -- a NONLINEAR problem, iterates
    # resident on the device, the model's functions evaluated there by the example's device model (torch operations: the model
    # is the caller's code, as Pyomo's is for the reference), everything else the library's kernels
    ip_loop_burgers = None
    if not args.no_ip_loop_dynamic and not args.no_ip_loop and args.workload in ('C3', 'C4') and not args.blocks and world == 1:
        try:
            from parapint_amd.algorithms.device_interior_point import ip_solve_device
            from parapint_amd.algorithms.interior_point import IPOptions, InteriorPointStatus
            from parapint_amd.examples import burgers
            Tb = args.ip_time_blocks
            best = None
            for rep in range(2):
                ipi = burgers.device_interface(50, Tb * 40, Tb)
                ipo = IPOptions()
                ipo.linalg.solver = HipSchurComplementLinearSolver({t: None for t in range(Tb)}, None, comm=comm, result_buffers=2)
                hist, ipst = [], {}
                sync_all()
                t0 = time.perf_counter()
                ip_status, ip_iters = ip_solve_device(ipi, ipo, history=hist, stats=ipst)
                sync_all()
                t_ip = time.perf_counter() - t0
                pg = max(ipi.pattern_groups, key=lambda g: len(g.members))
                cur = {'it_per_s': ip_iters / ipst['loop_s'], 'iterations': ip_iters,
                       'ms_per_iteration': 1e3 * ipst['loop_s'] / max(ip_iters, 1), 'loop_seconds': ipst['loop_s'],
                       'setup_seconds': t_ip - ipst['loop_s'], 'converged': ip_status == InteriorPointStatus.optimal,
                       'final_infeasibilities': list(hist[-1][:3]) if hist else None, 'objective': ipi.evaluate_objective(),
                       'time_blocks': Tb, 'variables_per_block': pg.n, 'block_dim': pg.nb, 'n_coupling': 2 * ipi.ncz,
                       'torch_ops_of_the_model': ipst.get('torch_ops')}
                if best is None or cur['it_per_s'] > best['it_per_s']:
                    best = cur
                del ipi, ipo
            ip_loop_burgers = best
        except Exception as exc:
            ip_loop_burgers = {'converged': False, 'error': '%s: %s' % (type(exc).__name__, exc)}

    # ---- what ONE rank of an 8-GPU run holds (strong scaling of the two sharded configurations): 128 blocks of C3, 512 of C5,
    # as separate one-GPU runs of this script -- the small-share regime is latency-bound and a target of its own
    shares = None
    if rank == 0 and world == 1 and args.workload == 'C3' and not args.blocks and not args.no_shares and not args.no_ip_loop:
        import subprocess
        shares = {}
        for key, extra in (('C3_128_blocks', ['--blocks', '128']),
                           ('C5_512_blocks', ['--workload', 'C5', '--blocks', '512', '--value-sets', '2', '--steps', '10', '--warmup', '2'])):
            try:
                cp = subprocess.run([sys.executable, os.path.abspath(__file__), '--no-cpu-baseline', '--no-boundary', '--no-ip-loop',
                                     '--no-shares'] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
                d = json.loads(cp.stdout.decode().strip().splitlines()[-1])
                shares[key] = {'ms_per_step': d['ms_per_step'], 'ms_per_step_no_prefetch': d['ms_per_step_no_prefetch'],
                               'ms_per_step_unchecked': d['solution_check']['ms_per_step_unchecked'], 'correct': d['correct'],
                               'phases_ms': {k: v['ms_per_step'] for k, v in d['phases'].items()},
                               'kernel_launches_per_step': d['kernel_launches_per_step']}
            except Exception as exc:
                shares[key] = {'error': '%s: %s' % (type(exc).__name__, exc)}
    if rank == 0:
        launches = sum(p['launches_per_step'] for p in phases.values()) if phases else None
        out = {
            'metric': METRIC, 'value': value, 'unit': 'it/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': args.scaling,
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': describe + '; per step 1 do_numeric_factorization + 1 do_back_solve (every back-solve '
                                   'checked on the device before it returns: value_unchecked is the step without) through the '
                                   'LinearSolverInterface methods on device-resident containers, fresh values each step '
                                   '(%d value sets in HBM cycled, every Hessian entry of every block differs)' % nsets,
                       'blocks_per_gpu': B, 'world_size': world, 'collective_backend': backend,
                       'parallelism': 'blocks round-robin over %d rank(s); all-reduce of [S | status | inertia] and '
                                      'of r_s' % world},
            'median_ms_per_step': median_ms,
            # `value`: the step as this package's own loops (ip_solve, ip_solve_device) run it -- the right-hand side is
            # announced (solver.prefetch_forward) before the factorisation; `value_no_prefetch`: the same K steps in the
            # reference's plain call order (interior_point.py:553-566), what an unaware caller sees
            'value_no_prefetch': value_no_prefetch, 'ms_per_step_no_prefetch': ms_no_prefetch,
            'value_unchecked': solution_check['value_unchecked'], 'solution_check': solution_check,
            'shares': shares,
            'rccl_ranks': rccl_ranks,      # > 0: the all-reduces were enqueued by the library (the default for >= 2 RCCL ranks)
            'collective_us': collective_us,             # per rank: the two data-path all-reduces by themselves (HIP events)
            # SURVEY 8(d) to the letter: the same step through HOST containers (SciPy COO blocks in, host vectors out;
            # staging, H2D and D2H inside) -- the rate a caller with the reference's unchanged interfaces sees
            'value_boundary': (boundary or {}).get('it_per_s'),
            'value_boundary_constant_declared': (declared or {}).get('it_per_s'),
            'value_boundary_flat_values': (flat_plain or {}).get('it_per_s'),
            'ip_loop': ip_loop,
            'ip_loop_dynamic': ip_loop_dynamic,
            'ip_loop_burgers': ip_loop_burgers,
            'roofline': roofline,
            'dense_phase': dense_phase,
            'cpu_baseline': cpu_baseline,
            'correct': bool(ok),
            'residual': resid, 'residual_boundary_host': resid_boundary,
            'inertia': list(inertia), 'expected_inertia': list(expected_inertia),
            'phases': phases, 'kernel_launches_per_step': launches,
            'plan': dict({k: st[k] for k in ('n', 'n_pivots', 'n_2x2', 'n_levels', 'nnz_L', 'u_doubles', 'factor_fma',
                                             'schur_fma', 'factor_tasks', 'canonical_entries', 'raw_entries')}, **ex),
            'survey_bytes_per_block': sb, 'build_bytes_per_block': bb,
            'symbolic_s': t_symbolic,
            'boundary_host': boundary, 'boundary_host_constant_declared': declared, 'boundary_host_flat_values': flat_plain,
            'device_only': device_only,
            'value_storage_bytes': {'device_resident_path': mem_now, 'with_host_input_and_output_copies': mem_max},
            'bcr_block_paths': bcr_paths,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        sys.exit(1)


if __name__ == '__main__':
    main()
