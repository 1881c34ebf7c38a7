"""The time-block (dynamic) Schur-complement interface (parapint/interfaces/schur_complement/sc_ip_interface.py:13-1026,
mpi_sc_ip_interface.py:32-270) through the restated interior-point loop: against the same
problem stated as one QP, over the oracle's solver classes, over the product's solver class (numpy engine on the CPU, the
HIP library on the device), on two ranks."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from parapint_amd.examples import dynamics_qp as dq

ARGS = dict(nfe_per_block=4, n_states=8, n_controls=2)


def _oracle_solver(blocks):
    from oracle.schur_complement import SchurComplementLinearSolver as OracleSC
    from oracle.subsolvers import ScipyInterface as OracleScipy
    return OracleSC({i: OracleScipy(compute_inertia=True) for i in blocks}, OracleScipy(compute_inertia=True))


def _monolithic(T, args):
    """The whole horizon as ONE QuadraticProgram through the same loop (one 'scenario'; its first variable doubles as
    the coupling variable the stochastic interface asks for)."""
    from parapint_amd.algorithms.interior_point import IPOptions, InteriorPointStatus, ip_solve
    from parapint_amd.interfaces.schur_complement.sc_ip_interface import StochasticSchurComplementInteriorPointInterface
    qp, off = dq.monolithic_qp(args, 0.0, 1.0, T)
    mono = StochasticSchurComplementInteriorPointInterface([qp], [[0]])
    opt = IPOptions()
    opt.linalg.solver = _oracle_solver([0])
    assert ip_solve(mono, opt) == InteriorPointStatus.optimal
    return mono, off


def _same_solution(it, mono, off, T, tol):
    assert abs(it.evaluate_objective() - mono.evaluate_objective()) <= tol
    x = mono.get_primals().get_block(0)
    for t in it.local_block_indices:
        assert np.abs(it.get_primals().get_block(t) - x[off[t]:off[t + 1]]).max() <= tol
    # the coupling states are the end states of the blocks before them
    z = np.asarray(it.get_primals().get_block(T))
    ns = it.num_states
    for t in range(T - 1):
        if t in it.local_block_indices:
            assert np.abs(z[ns * t:ns * (t + 1)] - it.get_primals().get_block(t)[it.ys(it.nfe)]).max() <= tol


def test_dynamic_interface_over_the_oracle_solver_matches_the_monolithic_problem():
    T = 4
    it = dq.main(_oracle_solver(range(T)), 0.0, 1.0, T, **ARGS)
    mono, off = _monolithic(T, ARGS)
    _same_solution(it, mono, off, T, 1e-7)
    u = np.concatenate([it.get_primals().get_block(t)[it.us(0)[0]:] for t in range(T)])
    assert np.isclose(u.max(), 1.5, atol=1e-5) and (np.abs(u) <= 1.5).all()       # the control bound is active


def test_dynamic_kkt_layout():
    """sc_ip_interface.py:274-357: sizes, symmetry, where the links sit, a block-banded Schur complement; and the
    regularisation hooks (:903-933) touch the diagonals they name."""
    T, ns = 4, ARGS['n_states']
    it = dq.DiffusionControl(0.0, 1.0, T, **ARGS)
    ncz = ns * (T - 1)
    assert it.n_eq_constraints() == sum(it.scenario_interface(t).n_eq_constraints() for t in range(T)) + 2 * ncz
    kkt = it.evaluate_primal_dual_kkt_matrix()
    M = kkt.tocoo().toarray()
    assert np.array_equal(M, M.T)
    dims = [kkt.get_block(t, t).shape[0] for t in range(T)]
    inner = [sum(getattr(it.scenario_interface(t), f)() for f in ('n_primals', 'n_eq_constraints')) +
             2 * it.scenario_interface(t).n_ineq_constraints() for t in range(T)]
    assert dims == [inner[0]] + [n + ns for n in inner[1:]] and kkt.get_block(T, T).shape == (2 * ncz, 2 * ncz)
    for t in range(T):
        A = kkt.get_block(T, t).tocoo().toarray()
        rows = np.flatnonzero(np.abs(A).sum(axis=1))
        want = ([] if t == T - 1 else list(ns * t + np.arange(ns))) + ([] if t == 0 else list(ncz + ns * (t - 1) + np.arange(ns)))
        assert list(rows) == want
        if t < T - 1:                                           # forward link: +1 on the end states
            assert np.array_equal(A[ns * t:ns * (t + 1)][:, it.ys(it.nfe)], np.eye(ns))
        if t > 0:                                               # -I on the multipliers of the backward link
            assert np.array_equal(A[ncz + ns * (t - 1):ncz + ns * t][:, inner[t]:], -np.eye(ns))
    Q = kkt.get_block(T, T).tocoo().toarray()
    assert np.array_equal(Q, np.block([[np.zeros((ncz, ncz)), -np.eye(ncz)], [-np.eye(ncz), np.zeros((ncz, ncz))]]))
    # rhs and solution round trip: the Newton step of the whole system solved densely goes back where it belongs
    it.set_barrier_parameter(0.1)
    rhs = it.evaluate_primal_dual_kkt_rhs()
    assert rhs.get_block(T).size == 2 * ncz
    k2 = it.regularize_equality_gradient(kkt, -1e-3, copy_kkt=True)
    k2 = it.regularize_hessian(k2, 1e-2, copy_kkt=False)
    D = k2.tocoo().toarray() - M
    assert np.array_equal(D, np.diag(np.diag(D)))
    d, o = np.diag(D), 0
    for t in range(T):
        nlp = it.scenario_interface(t)
        n, mi, me = nlp.n_primals(), nlp.n_ineq_constraints(), nlp.n_eq_constraints()
        nb = ns if t else 0
        assert np.allclose(d[o:o + n], 1e-2) and np.allclose(d[o + n:o + n + mi], 0) and \
            np.allclose(d[o + n + mi:o + n + mi + me], -1e-3) and np.allclose(d[o + n + 2 * mi + me:o + dims[t]], -1e-3)
        assert d[o + n + 2 * mi + me:o + dims[t]].size == nb
        o += dims[t]
    assert np.allclose(d[o:o + ncz], -1e-3) and np.allclose(d[o + ncz:], 1e-2)
    assert np.array_equal(kkt.tocoo().toarray(), M)             # copy_kkt=True left the original alone


def test_dynamic_loop_over_the_product_solver_on_the_cpu_engine():
    from hostsim_engine import HostSimEngine
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
    T = 6
    solver = HipSchurComplementLinearSolver({t: None for t in range(T)}, None, comm=SerialComm(), engine=HostSimEngine())
    it = dq.main(solver, 0.0, 1.0, T, **ARGS)
    assert sorted(len(g.blocks) for g in solver._groups) == [1, 1, T - 2]     # first, inner and last time blocks
    mono, off = _monolithic(T, ARGS)
    _same_solution(it, mono, off, T, 1e-7)


def test_more_ranks_than_time_blocks_is_refused():
    class Two(object):
        rank, size = 0, 3
    with pytest.raises(ValueError, match='more processes than time blocks'):
        dq.DiffusionControl(0.0, 1.0, 2, comm=Two(), **ARGS)


def _two_rank_run(*args):
    here = os.path.dirname(os.path.abspath(__file__))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(here, 'dynamic_multirank_worker.py')] + list(args)
    env = dict(os.environ)
    env['OMP_NUM_THREADS'] = '1'
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = out.stdout.decode()
    assert out.returncode == 0, text[-4000:]
    assert 'rank 0 ok' in text and 'rank 1 ok' in text


def test_two_rank_dynamic_loop():
    _two_rank_run()


@pytest.mark.gpu
def test_two_rank_dynamic_loop_on_the_device():
    _two_rank_run('--gpu')


@pytest.mark.gpu
def test_dynamic_loop_over_the_hip_solver():
    """24 time blocks x 20 states: the block-tridiagonal S path of the library inside a real interior-point loop."""
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
    T, args = 24, dict(nfe_per_block=4, n_states=20, n_controls=3, nu=0.01)
    solver = HipSchurComplementLinearSolver({t: None for t in range(T)}, None, comm=SerialComm())
    it = dq.main(solver, 0.0, 1.0, T, **args)
    assert sorted(len(g.blocks) for g in solver._groups) == [1, 1, T - 2]
    # the checker: the same loop over the oracle's solver classes (the decomposition itself is pinned against the
    # monolithic problem in the CPU tests; its dense-eigenvalue inertia would take minutes at this size)
    ref = dq.main(_oracle_solver(range(T)), 0.0, 1.0, T, **args)
    assert abs(it.evaluate_objective() - ref.evaluate_objective()) <= 1e-7
    for t in range(T + 1):
        assert np.abs(np.asarray(it.get_primals().get_block(t)) - np.asarray(ref.get_primals().get_block(t))).max() <= 1e-6


# ---- the device-resident producer for time blocks (DeviceDynamicQPInterface; include/parapint_hip.h: mapped groups) -----
def _time_blocks(T, args):
    return dq.DiffusionControl.time_blocks(0.0, 1.0, T, **args)


def _device_loop(blocks, engine, comm=None, local=None):
    from parapint_amd.algorithms.device_interior_point import ip_solve_device
    from parapint_amd.algorithms.interior_point import IPOptions, InteriorPointStatus
    from parapint_amd.interfaces.schur_complement.device_sc_ip_interface import DeviceDynamicQPInterface
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
    it = DeviceDynamicQPInterface(blocks, comm=comm)
    opt = IPOptions()
    opt.linalg.solver = HipSchurComplementLinearSolver({t: None for t in it.local}, None, comm=comm or SerialComm(),
                                                       engine=engine, result_buffers=0 if engine is not None else 2)
    hist, stats = [], {}
    status, iters = ip_solve_device(it, opt, history=hist, stats=stats)
    assert status == InteriorPointStatus.optimal
    stats['solver'] = opt.linalg.solver
    return it, hist, stats


def _host_history(T, args, solver):
    """Rows of the host loop's log: iteration, objective, primal / dual / complementarity infeasibility, barrier."""
    import logging
    import re
    rows = []

    class Cap(logging.Handler):
        def emit(self, record):
            m = re.match(r'^(\d+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)', record.getMessage())
            if m:
                rows.append([float(v) for v in m.groups()])
    log = logging.getLogger('parapint_amd.algorithms.interior_point')
    cap, old = Cap(), log.level
    log.addHandler(cap)
    log.setLevel(logging.INFO)
    try:
        it = dq.main(solver, 0.0, 1.0, T, **args)
    finally:
        log.removeHandler(cap)
        log.setLevel(old)
    return it, rows


def _same_iterations(rows, hist, primal_noise=2e-9):
    """The host loop prints 3 significant digits; below 1e-9 the measures are rounding noise of two summation orders.
    primal_noise: with a singular KKT matrix regularised by 1e-8 the residual of the constraints after a step is the
    rounding error of a solve at condition 1e8 and more -- two orders of summation give different noise there."""
    assert len(rows) == len(hist)
    for r, h in zip(rows, hist):
        for k, (a, b) in enumerate(zip(r[2:6], h[:4])):
            assert abs(a - b) <= 6e-3 * max(abs(a), abs(b)) + (primal_noise if k == 0 else 2e-9), (r, h)


def _same_point(it, host, T, tol):
    z = np.asarray(host.get_primals().get_block(T))
    assert np.abs(it.coupling_states() - z).max() <= tol
    for t in it.local:
        assert np.abs(it.scenario_primals(t) - host.get_primals().get_block(t)).max() <= tol


def _product_solver_on_cpu(T):
    from hostsim_engine import HostSimEngine
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
    return HipSchurComplementLinearSolver({t: None for t in range(T)}, None, comm=SerialComm(), engine=HostSimEngine())


def test_device_dynamic_loop_on_cpu_engines_matches_the_host_loop():
    """Producer + loop on the numpy restatement of the kernels: iteration by iteration the measures of the host loop over
    the same solver class, the same point at the end, three pattern groups (first, inner, last time blocks)."""
    from hostsim_engine import HostSimDeviceEngine
    T = 6
    it, hist, _ = _device_loop(_time_blocks(T, ARGS), HostSimDeviceEngine())
    assert sorted(len(pg.members) for pg in it.pattern_groups) == [1, 1, T - 2]
    assert sorted((pg.nfs, pg.nfw) for pg in it.pattern_groups) == [(0, 8), (8, 0), (8, 8)]
    host, rows = _host_history(T, ARGS, _product_solver_on_cpu(T))
    _same_iterations(rows, hist)
    _same_point(it, host, T, 1e-7)
    mono, off = _monolithic(T, ARGS)
    assert abs(it.evaluate_objective() - mono.evaluate_objective()) <= 1e-7


def test_device_dynamic_loop_with_two_time_blocks_and_refused_inputs():
    """Two time blocks: no inner pattern group, one coupling state set; one time block, or more ranks than blocks: refused."""
    from hostsim_engine import HostSimDeviceEngine
    from parapint_amd.interfaces.schur_complement.device_sc_ip_interface import DeviceDynamicQPInterface
    it, hist, _ = _device_loop(_time_blocks(2, ARGS), HostSimDeviceEngine())
    assert sorted((pg.nfs, pg.nfw) for pg in it.pattern_groups) == [(0, 8), (8, 0)] and it.ncz == 8
    host, rows = _host_history(2, ARGS, _product_solver_on_cpu(2))
    _same_iterations(rows, hist)
    _same_point(it, host, 2, 1e-7)
    with pytest.raises(ValueError, match='at least two time blocks'):
        DeviceDynamicQPInterface(_time_blocks(2, ARGS)[:1])

    class Three(object):
        rank, size = 0, 3
    with pytest.raises(ValueError, match='more processes than time blocks'):
        DeviceDynamicQPInterface(_time_blocks(2, ARGS), comm=Three())


def test_device_dynamic_loop_regularises_like_the_host_loop():
    """A rank-deficient Jacobian in every time block: every iteration goes through the inertia-correction retries -- on the
    device as diagonal shifts of the resident values, with the rows of the forward multipliers in the coupling block
    classed as constraint rows (sc_ip_interface.py:903-933)."""
    from hostsim_engine import HostSimDeviceEngine
    T, args = 5, dict(ARGS, duplicate_constraint=True)
    it, hist, _ = _device_loop(_time_blocks(T, args), HostSimDeviceEngine())
    assert it.solver.diagonal_shift_refactorizations >= len(hist) - 1
    host, rows = _host_history(T, args, _product_solver_on_cpu(T))
    assert all(r[9] > 0 for r in rows[1:])                       # the Reg column: every iteration was regularised
    _same_iterations(rows, hist, primal_noise=1e-6)
    _same_point(it, host, T, 1e-6)


def test_device_dynamic_loop_with_a_block_tridiagonal_coupling_block():
    """40 time blocks x 14 states: the coupling block (1092 rows) is beyond the dense limit, S is kept block tridiagonal
    and solved by cyclic reduction; r_s / x_s pass through the solver's permutation of the coupling variables."""
    from hostsim_engine import HostSimDeviceEngine
    T, args = 40, dict(nfe_per_block=2, n_states=14, n_controls=2, nu=0.02)
    it, hist, _ = _device_loop(_time_blocks(T, args), HostSimDeviceEngine())
    assert it.solver._btd is not None and 2 * it.ncz == 1092
    host, rows = _host_history(T, args, _product_solver_on_cpu(T))
    _same_iterations(rows, hist)
    _same_point(it, host, T, 1e-7)


def test_two_rank_device_dynamic_loop():
    _two_rank_run('--device-producer')


@pytest.mark.gpu
def test_two_rank_device_dynamic_loop_on_the_device():
    _two_rank_run('--device-producer', '--gpu')


@pytest.mark.gpu
@pytest.mark.parametrize('T,args', [(6, ARGS), (24, dict(nfe_per_block=4, n_states=20, n_controls=3, nu=0.01)),
                                    (5, dict(ARGS, duplicate_constraint=True)),
                                    (3, dict(nfe_per_block=1, n_states=1, n_controls=1)),          # no inequality rows
                                    (7, dict(nfe_per_block=2, n_states=3, n_controls=5)),
                                    (130, dict(nfe_per_block=2, n_states=4, n_controls=1, nu=0.01))])   # 128 inner lanes: two per lane
def test_device_dynamic_loop_kernels_match_their_numpy_restatement(T, args):
    """The same loop on the HIP kernels and on their numpy restatement: the same iterations, measures to rounding (the
    two solvers sum in different orders), no torch operator in an iteration."""
    from hostsim_engine import HostSimDeviceEngine
    blocks = _time_blocks(T, args)
    it, hist, stats = _device_loop(blocks, None)
    ref, ref_hist, _ = _device_loop(blocks, HostSimDeviceEngine())
    regularised = bool(args.get('duplicate_constraint'))
    # (a retry of the inertia-correction loop sends the shifted corner of the coupling block from the host: the only
    # torch operators of the loop)
    # (one-off: the index tensor of the block-tridiagonal permutation goes to the device through torch)
    assert len(hist) == len(ref_hist) and (stats['torch_ops'] <= 6 or regularised)
    for a, b in zip(hist, ref_hist):
        assert np.allclose(a[1:6], b[1:6], rtol=1e-6, atol=2e-9), (a, b)
        assert abs(a[0] - b[0]) <= 1e-6 * abs(b[0]) + (1e-6 if regularised else 2e-9), (a, b)     # see _same_iterations
    assert np.abs(it.coupling_states() - ref.coupling_states()).max() <= 1e-7
    for t in range(T):
        assert np.abs(it.scenario_primals(t) - ref.scenario_primals(t)).max() <= 1e-7


@pytest.mark.gpu
def test_device_dynamic_loop_at_a_longer_horizon_against_the_host_producer():
    """96 time blocks x 30 states (coupling block 5700, block-tridiagonal S on the device): the device-resident producer
    against the host producer (the restated reference interface, pinned to the oracle's solver classes and to the
    monolithic problem at small sizes above) over the same HIP solver class, iteration by iteration."""
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
    T, args = 96, dict(nfe_per_block=4, n_states=30, n_controls=3, nu=0.005)
    it, hist, stats = _device_loop(_time_blocks(T, args), None)
    assert stats['torch_ops'] <= 6, stats['torch_op_names']      # (set-up of the permutation; none per iteration)
    # the diagonal blocks of this S carry next to nothing on their diagonals (link duals AND coupling states of a quadratic
    # program): with the pair rotations of k_bcr_ldl_inverse every one of them still takes the unpivoted matrix-core path in
    # the last factorisation of the run (round 3: none of them did)
    fast, pivoted = stats['solver']._eng.bcr_block_paths()
    assert (fast, pivoted) == (T - 1, 0), (fast, pivoted)
    host, rows = _host_history(T, args, HipSchurComplementLinearSolver({t: None for t in range(T)}, None, comm=SerialComm()))
    _same_iterations(rows, hist)
    _same_point(it, host, T, 1e-6)
    assert abs(it.evaluate_objective() - host.evaluate_objective()) <= 1e-8


def _mapped_step_problem(seed, ns=5, inner=70, n=33, me=9, mi=4):
    """Random iterates for three mapped groups laid out like time blocks (first: forward link only, `inner` blocks with
    both links, last: backward link only) over ncz = ns * (inner + 1) coupling states; host arrays [row][instance]."""
    from test_device_ip import random_stochastic_qp
    from parapint_amd.interfaces.schur_complement.device_sc_ip_interface import _PatternGroup
    rng = np.random.default_rng(seed)
    T = inner + 2
    ncz = ns * (T - 1)
    descs = []
    for gi, (lanes, has_b, has_f) in enumerate((([0], False, True), (list(range(1, T - 1)), True, True), ([T - 1], True, False))):
        qps, fs = random_stochastic_qp(1, n=n, n_fs=ns, n_eq=me, n_ineq=mi, seed=seed + gi)
        fsb = np.asarray(fs[0]) if has_b else np.zeros(0, dtype=np.int64)
        ff = (np.asarray(fs[0]) + 7) % n if has_f else np.zeros(0, dtype=np.int64)
        pg = _PatternGroup(qps[0], fsb, ff, mapped=True)
        prog, terms = pg.row_programs()
        B = len(lanes)
        bpad = -(-B // 64) * 64
        nb, nfw = pg.nb, pg.nfw
        W = rng.uniform(0.5, 2.0, size=(nb + 2 * n + 2 * mi + nfw, bpad))
        W[n + mi:nb] = rng.normal(size=(nb - n - mi, bpad))
        W[nb + 2 * n + 2 * mi:] = rng.normal(size=(nfw, bpad))                 # copies of the forward multipliers
        lo = W[:n + mi] - rng.uniform(0.1, 1.0, size=(n + mi, bpad))
        hi = W[:n + mi] + rng.uniform(0.1, 1.0, size=(n + mi, bpad))
        lo[rng.random(lo.shape) < 0.3] = -np.inf
        hi[rng.random(hi.shape) < 0.3] = np.inf
        bounds = np.concatenate([lo[:n], hi[:n], lo[n:], hi[n:]])
        e = nb + 2 * n + 2 * mi
        zl = np.concatenate([W[nb:nb + n], W[nb + 2 * n:nb + 2 * n + mi]])
        zu = np.concatenate([W[nb + n:nb + 2 * n], W[nb + 2 * n + mi:e]])
        zl[~np.isfinite(lo)] = 0.0
        zu[~np.isfinite(hi)] = 0.0
        W[nb:nb + n], W[nb + 2 * n:nb + 2 * n + mi] = zl[:n], zl[n:]
        W[nb + n:nb + 2 * n], W[nb + 2 * n + mi:e] = zu[:n], zu[n:]
        src = np.zeros((pg.nsrc, bpad))
        src[:pg.off[3]] = rng.normal(size=(pg.off[3], bpad))
        full = lanes + [lanes[0]] * (bpad - B)
        zoff = np.zeros((2, bpad), dtype=np.int32)
        zoff[0] = [ns * (t - 1) if t > 0 else 0 for t in full]
        zoff[1] = [ns * t if t < T - 1 else 0 for t in full]
        descs.append(dict(n=n, mi=mi, me=me, nfs=pg.nfs, nfw=nfw, ncz=ncz, zoff=zoff, batch=B, bpad=bpad,
                          src_dp=int(pg.off[3]), src_ds=int(pg.off[4]), W=W, bounds=bounds,
                          data=rng.normal(size=(n + me, bpad)), src=src, G=np.zeros((n, bpad)), rhs=np.zeros((nb, bpad)),
                          prog=prog, terms=terms, delta=rng.normal(size=(nb, bpad))))
    return descs, rng.normal(size=ncz), rng.normal(size=2 * ncz), ncz


@pytest.mark.gpu
def test_mapped_step_kernels_match_their_numpy_restatement():
    """The step kernels on mapped groups (per-instance coupling offsets, forward multipliers kept per instance, coupling
    right-hand side scattered) against the numpy restatement: elementwise results, step lengths, max-norms and the
    scattered coupling rows bit for bit (every coupling entry has at most two addends), sums to rounding."""
    from hostsim_ip_ops import HostSimIpOps
    from test_device_ip import _run_step_sequence
    from parapint_amd.linalg.hip_schur_complement import HipEngine
    descs, z, dz, ncz = _mapped_step_problem(23)
    mu, tau = 0.1, 0.9
    copy = [dict(d, **{k: v.copy() for k, v in d.items() if isinstance(v, np.ndarray)}) for d in descs]
    ref = _run_step_sequence(HostSimIpOps(), copy, z, dz, mu, tau, ncoup=2 * ncz, dual_from=ncz)
    got = _run_step_sequence(HipEngine().ip_ops(), descs, z, dz, mu, tau, ncoup=2 * ncz, dual_from=ncz)
    assert np.array_equal(got['alpha'], ref['alpha']) and 0.0 < ref['alpha'].min() < 1.0
    for key in ('rhs0', 'src0', 'W1', 'G1', 'rhs1', 'src1'):
        for gi, (a, b) in enumerate(zip(got[key], ref[key])):
            B = descs[gi]['batch']
            assert np.array_equal(a[:, :B], b[:, :B]), (key, gi, np.abs(a[:, :B] - b[:, :B]).max())
    assert np.array_equal(got['z1'], ref['z1'])
    assert np.array_equal(got['rc1'], ref['rc1']) and np.array_equal(got['v0'][8:], ref['v0'][8:])     # scattered rows
    assert np.abs(ref['rc1'][:ncz]).min() > 0 and np.abs(ref['rc1'][ncz:]).min() > 0                  # every entry written
    for key in ('mail0', 'mail1'):
        a, b = got[key], ref[key]
        assert np.array_equal(a[[0, 1, 2, 3, 7, 8]], b[[0, 1, 2, 3, 7, 8]]), (key, a, b)                # max / min: exact
        assert np.allclose(a[[4, 5, 6]], b[[4, 5, 6]], rtol=1e-13, atol=0.0), (key, a, b)
    # offsets outside the coupling states are refused before any kernel sees them
    bad = dict(descs[1], zoff=descs[1]['zoff'].copy())
    bad['zoff'][1, 3] = ncz - 2
    ops = HipEngine().ip_ops()
    with pytest.raises(ValueError, match='outside'):
        ops.prepare([{k: (ops.from_host(v) if isinstance(v, np.ndarray) else v) for k, v in bad.items() if k != 'delta'}])
