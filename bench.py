#!/usr/bin/env python
"""Benchmark of the hot path: interior-point iterations / second on the synthetic stochastic KKT.

One step = one IP-iteration-equivalent of linear algebra (reference call
sites parapint/algorithms/interior_point.py:553-567): ONE numeric factorisation on fresh values
(same pattern) + ONE back-solve, including the status/inertia read-back the IP loop needs.
Inputs (K_i values of every block for that iteration, right-hand sides) are resident in HBM when
the timed region starts; the PCIe-inclusive rate through the LinearSolverInterface boundary is
measured separately and reported in `boundary`.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Workload: BASELINE.json configs[2] (the one the metric is quoted on): 1024 scenarios x
(n_q=1000, n_y=4000 -> 5000 primal vars, block dim 9200), 200 coupling variables, sharded
round-robin over the ranks (strong scaling: the total is fixed).
"""
import argparse
import json
import os
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


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--blocks', type=int, default=1024)
    ap.add_argument('--n-q', type=int, default=1000)
    ap.add_argument('--m', type=int, default=4)
    ap.add_argument('--n-theta', type=int, default=200)
    ap.add_argument('--value-sets', type=int, default=6, help='distinct pre-staged value sets cycled through')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-boundary', action='store_true')
    ap.add_argument('--profile-steps', type=int, default=5)
    ap.add_argument('--sn-wmax', type=int, default=0, help='supernode width cap (0: library default)')
    ap.add_argument('--sn-tol', type=int, default=-1, help='padded rows tolerated when merging (-1: default)')
    ap.add_argument('--splits', type=int, default=0, help='instance splits on separate streams (0: library default)')
    return ap.parse_args()


def survey_bytes_per_block(z_K, z_L, n_i, n_c, z_A):
    """Algorithmic HBM bytes per block and iteration, SURVEY.md section 8(d) (tight Schur model B_sc')."""
    b_fac = 12 * z_K + 12 * z_L
    b_sc = 2 * 12 * z_L + 8 * n_c * n_c
    b_bs = 2 * (2 * 12 * z_L + 3 * 8 * n_i) + 2 * 12 * z_A
    return {'factor': b_fac, 'schur': b_sc, 'back_solve': b_bs, 'total': b_fac + b_sc + b_bs}


def main():
    args = parse_args()
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run for --gpus > 1')
    N, n_q, m, n_t = args.blocks, args.n_q, args.m, args.n_theta

    # ---- CPU baseline first: it forks worker processes, so it runs before the GPU is touched
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cpu_baseline as cb
        cpu_baseline = cb.run(N, n_q, m, n_t, blocks_per_worker=max(1, min(64, N // 16)))   # ~10 s on 16 cores at C3

    import torch
    import torch.distributed as dist
    from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT, distribute_blocks
    from parapint_amd.linalg.comm import SerialComm, TorchComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver

    # Rehearsal switch (not the measured configuration): PP_BENCH_REHEARSAL=gloo lets several ranks share the GPUs
    # that are present and exchange through gloo (host-staged all-reduce), to exercise the N > 1 code path on a
    # one-GPU box.  The driver's multi-GPU runs use one GPU per rank and RCCL.
    rehearsal = os.environ.get('PP_BENCH_REHEARSAL', '')
    if rehearsal:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if rehearsal:
            dist.init_process_group(rehearsal, rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        comm = TorchComm()
    else:
        comm = SerialComm()
    dev = torch.device('cuda', local_rank)

    local = distribute_blocks(N, rank, world)
    model = SyntheticKKT(N, n_q, m, n_t, local_blocks=local)
    B = len(local)
    solver = HipSchurComplementLinearSolver({i: None for i in local}, None, comm=comm)
    eng = solver._eng
    if args.sn_wmax > 0 or args.sn_tol >= 0:
        eng.set_supernodes(args.sn_wmax, args.sn_tol)
    if args.splits > 0:
        eng.ns.check(eng.lib.pp_set_instance_splits(eng.ns.h, args.splits), 'pp_set_instance_splits')

    # ---- through the LinearSolverInterface boundary (host buffers in, host buffers out)
    kkt = model.build_kkt(comm=comm, iteration=0)
    rhs = model.build_rhs(comm=comm)
    t0 = time.perf_counter()
    solver.do_symbolic_factorization(kkt)
    t_symbolic = time.perf_counter() - t0
    st = solver.plan_stats[0]
    boundary = None
    t0 = time.perf_counter()
    solver.do_numeric_factorization(kkt)
    x = solver.do_back_solve(rhs)
    t_first = time.perf_counter() - t0
    if not args.no_boundary:
        ts = []
        for it in (1, 2, 3):
            kkt_it = model.build_kkt(comm=comm, iteration=it)
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            solver.do_numeric_factorization(kkt_it)
            x = solver.do_back_solve(rhs)
            ts.append(time.perf_counter() - t0)
            kkt = kkt_it
        boundary = {'it_per_s': 1.0 / float(np.median(ts)), 'ms_per_iteration': 1e3 * float(np.median(ts)),
                    'note': 'host COO blocks in, host vectors out (staging + PCIe H2D/D2H included)'}

    # ---- correctness gate on the last boundary solve: scaled residual, inertia
    def residual_check(x_blocks, xc, iteration):
        A = model.border_matrix().tocsr()
        worst = 0.0
        rc_local = np.zeros(n_t)
        for slot, ndx in enumerate(local):
            K = model.block_matrix(ndx, iteration).tocsr()
            r = model.block_rhs(ndx)
            xi = x_blocks[slot]
            res = K @ xi + A.T @ xc - r
            scale = abs(K).sum(axis=1).max() * max(np.abs(xi).max(), np.abs(xc).max()) + np.abs(r).max()
            worst = max(worst, float(np.abs(res).max() / scale))
            rc_local += A @ xi
        rc = comm.allreduce_sum(rc_local) if world > 1 else rc_local
        worst = max(worst, float(np.abs(rc).max() / (np.abs(xc).max() * B + 1e-300)))
        if world > 1:
            worst = float(comm.allreduce_max(np.array([worst]))[0])
        return worst

    last_it = 0 if args.no_boundary else 3
    xb = [np.asarray(x.get_block(ndx)) for ndx in local]
    resid_boundary = residual_check(xb, np.asarray(x.get_block(N)), last_it)
    inertia = solver.get_inertia()
    expected_inertia = (N * (model.n_y + n_q) + n_t, N * (model.n_y + n_t), 0)
    ok = resid_boundary <= 1e-8 and tuple(inertia) == expected_inertia

    # ---- device-resident steps: value sets and right-hand sides staged in HBM beforehand
    nsets = max(2, min(args.value_sets, args.steps + args.warmup))
    base = np.concatenate([model._base, model.border_matrix().data])
    raw_sets = []
    for s in range(nsets):
        vals = np.tile(base, (B, 1))
        for slot, ndx in enumerate(local):
            vals[slot, :model.n_y] = 2.0 + np.random.default_rng(10_000 * (100 + s) + ndx).uniform(0.0, 0.5)
        raw_sets.append(torch.from_numpy(vals).to(dev))
    rhs_dev = torch.from_numpy(np.stack([model.block_rhs(ndx) for ndx in local])).to(dev)
    lib, h = eng.lib, eng.ns.h
    eng.ns.check(lib.pp_bind_rhs_buffer(h, 0, rhs_dev.data_ptr()), 'pp_bind_rhs_buffer')

    def step(k):
        eng.ns.check(lib.pp_bind_raw_buffer(h, 0, raw_sets[k % nsets].data_ptr()), 'pp_bind_raw_buffer')
        eng.numeric_local()
        eng.allreduce_schur(comm)
        eng.factor_schur(None)
        status = eng.status()            # status + inertia read-back, as the IP loop needs every iteration
        eng.solve_forward()
        eng.allreduce_rs(comm)
        eng.solve_coupling(None)
        eng.solve_backward()
        return status

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for k in range(args.warmup):
        step(k)
    sync_all()
    t0 = time.perf_counter()
    for k in range(args.steps):
        status = step(args.warmup + k)
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        elapsed = float(comm.allreduce_max(np.array([elapsed]))[0])
    ms_per_step = 1e3 * elapsed / args.steps
    value = args.steps / elapsed

    # correctness of the last timed step (device path): download and check
    k_last = (args.warmup + args.steps - 1) % nsets
    xdev = np.zeros((B, model.block_dim))
    eng.download_solution(0, xdev)
    xc = eng.coupling_solution()
    A = model.border_matrix().tocsr()
    worst = 0.0
    rc_local = np.zeros(n_t)
    raw_last = raw_sets[k_last].cpu().numpy()
    from scipy.sparse import coo_matrix
    for slot, ndx in enumerate(local):
        K = coo_matrix((raw_last[slot, :model.nnz_per_block], (model._row, model._col)),
                       shape=(model.block_dim, model.block_dim)).tocsr()
        r = model.block_rhs(ndx)
        res = K @ xdev[slot] + A.T @ xc - r
        scale = abs(K).sum(axis=1).max() * max(np.abs(xdev[slot]).max(), np.abs(xc).max()) + np.abs(r).max()
        worst = max(worst, float(np.abs(res).max() / scale))
        rc_local += A @ xdev[slot]
    rc = comm.allreduce_sum(rc_local) if world > 1 else rc_local
    worst = max(worst, float(np.abs(rc).max() / (np.abs(xc).max() * B + 1e-300)))
    if world > 1:
        worst = float(comm.allreduce_max(np.array([worst]))[0])
    ok = ok and worst <= 1e-8 and status[0] == 0 and tuple(status[1:]) == expected_inertia

    # ---- per-phase device time (HIP events on the solver's stream), separate untimed pass
    import ctypes
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
    phases = {}
    for i, name in enumerate(PHASES):
        if calls[i] > 0:
            phases[name] = {'ms_per_step': float(ms[i] / args.profile_steps),
                            'launches_per_step': int(launches[i] // args.profile_steps)}

    # ---- roofline of the dominant kernel class (by device time)
    z_K = st['canonical_entries']
    z_L = st['u_doubles']
    sb = survey_bytes_per_block(z_K, z_L, st['n'], n_t, n_t)
    phase_bytes = {'assemble': 8.0 * (2 * st['raw_entries'] + z_K + z_L), 'factor_levels': float(sb['factor']),
                   'schur_tiles': float(sb['schur']), 'fwd_levels': sb['back_solve'] / 2.0,
                   'bwd_levels': sb['back_solve'] / 2.0}
    if not any(p in phase_bytes for p in phases):
        raise SystemExit('bench.py needs --profile-steps >= 1 for the roofline entry')
    dom = max((p for p in phases if p in phase_bytes), key=lambda p: phases[p]['ms_per_step'])
    dom_ms = phases[dom]['ms_per_step']
    dom_launches = max(1, phases[dom]['launches_per_step'])
    bytes_per_launch = phase_bytes[dom] * B / dom_launches
    achieved = bytes_per_launch / (dom_ms / dom_launches * 1e-3) / 1e9
    # HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
    # runs of this same command; profiles/pmc_traffic.json) -- valid for the build it was collected on
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
        ph = pmc['phases'].get(dom)
        if ph and ph['launches_per_step'] > 0 and world == 1 and N == 1024 and n_q == 1000:
            traffic = ph['hbm_bytes_per_step'] / ph['launches_per_step']
    except Exception:
        traffic = None
    roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                'algorithmic_bytes_per_launch': bytes_per_launch,
                'avg_launch_us': 1e3 * dom_ms / dom_launches,
                'whole_iteration': {'bytes': sb['total'] * B, 'GBps': sb['total'] * B / (ms_per_step * 1e-3) / 1e9,
                                    'frac': sb['total'] * B / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS}}

    # dense phase (factorisation of S, replicated on every rank): fp64 MFMA work, SURVEY 8d F_S = n_c^3/3 + 4 n_c^2
    dense_phase = None
    if 'dense_S' in phases and n_t > 0:
        f_s = n_t ** 3 / 3.0 + 4.0 * n_t ** 2
        tf = f_s / (phases['dense_S']['ms_per_step'] * 1e-3) / 1e12
        dense_phase = {'flops': f_s, 'ms': phases['dense_S']['ms_per_step'], 'achieved_TFLOPs': tf,
                       'peak_fp64_mfma_TFLOPs': FP64_MFMA_PEAK_TF, 'frac': tf / FP64_MFMA_PEAK_TF,
                       'note': 'one workgroup, latency-bound chain of n_c/16 panels; immaterial to the rate at n_c = 200'}

    if rank == 0:
        out = {
            'metric': METRIC, 'value': value, 'unit': 'it/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'C3: %d scenario blocks x (n_q=%d, n_y=%d: %d primal vars, block dim %d), '
                                   '%d coupling vars; 1 numeric factorisation + 1 back-solve per step, fresh '
                                   'values each step (%d device-resident value sets cycled)' %
                                   (N, n_q, model.n_y, n_q + model.n_y, model.block_dim, n_t, nsets),
                       'blocks_per_gpu': B, 'parallelism': 'blocks round-robin over %d rank(s); RCCL all-reduce of '
                                                           'S (+status) and r_s' % world},
            'roofline': roofline,
            'dense_phase': dense_phase,
            'cpu_baseline': cpu_baseline,
            'correct': bool(ok),
            'residual_device_path': worst, 'residual_boundary_path': resid_boundary,
            'inertia': list(inertia), 'expected_inertia': list(expected_inertia),
            'phases': phases,
            'plan': {k: st[k] for k in ('n', 'n_pivots', 'n_2x2', 'n_levels', 'nnz_L', 'u_doubles', 'factor_fma',
                                        'schur_fma', 'factor_tasks', 'canonical_entries', 'raw_entries')},
            'survey_bytes_per_block': sb,
            'symbolic_s': t_symbolic, 'first_numeric_plus_solve_s': t_first,
            'boundary': boundary,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        sys.exit(1)


if __name__ == '__main__':
    main()
