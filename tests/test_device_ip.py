"""The interior-point loop with device-resident iterates: the producer's value
map and right-hand side against the host interface (CPU), and -- on the GPU -- the farmer problem and a 256-scenario
stochastic QP through ``ip_solve_device``, iteration by iteration against the host loop of the same solver class."""
import logging
import re

import numpy as np
import pytest

from parapint_amd.examples import stochastic as ex
from parapint_amd.examples.stochastic_qp import random_stochastic_qp
from parapint_amd.interfaces.schur_complement.device_sc_ip_interface import DeviceStochasticQPInterface


def farmer_qps(extra=1):
    farmer = ex.Farmer(extra_scenarios=extra)
    qps, fs = [], []
    for s in farmer.scenarios:
        qp, acre = ex.create_scenario(farmer, s)
        qps.append(qp)
        fs.append(acre)
    return qps, fs


@pytest.mark.parametrize('problem', ['farmer', 'qp'])
def test_value_map_reproduces_the_host_kkt_matrix(problem):
    qps, fs = farmer_qps() if problem == 'farmer' else random_stochastic_qp(5, seed=3)
    it = DeviceStochasticQPInterface(qps, fs)
    dk = it.device_kkt_matrix()
    host_kkt = it.host.evaluate_primal_dual_kkt_matrix()      # (dk itself carries one scenario's blocks for every scenario of a pattern group)
    N = len(qps)
    for ndx in range(N):
        src, coef = it.group_of[ndx].value_map
        nlp = it.host.scenario_interface(ndx)
        dp, ds = nlp.barrier_diagonals()
        q = qps[ndx]
        source = np.concatenate([q.H.data, q.A_eq.data, q.A_ineq.data, dp, ds])
        vals = coef * np.where(src >= 0, source[np.maximum(src, 0)], 1.0)
        ref = np.concatenate([host_kkt.get_block(ndx, ndx).tocoo().data, host_kkt.get_block(N, ndx).tocoo().data])
        assert np.array_equal(vals, ref)
        # the pattern matrix of the symbolic phase: the same entries in the same order for every scenario
        K, A = dk.get_block(ndx, ndx).tocoo(), dk.get_block(N, ndx).tocoo()
        Kh, Ah = host_kkt.get_block(ndx, ndx).tocoo(), host_kkt.get_block(N, ndx).tocoo()
        assert np.array_equal(K.row, Kh.row) and np.array_equal(K.col, Kh.col)
        assert np.array_equal(A.row, Ah.row) and np.array_equal(A.col, Ah.col)
    assert dk.nsrc == it.nsrc == max(pg.nsrc for pg in it.pattern_groups) and len(src) == ref.size


def test_shifted_device_matrix_bookkeeping():
    qps, fs = random_stochastic_qp(3, seed=1)
    it = DeviceStochasticQPInterface(qps, fs)
    dk = it.device_kkt_matrix()
    k1 = dk.copy()
    assert k1.base is dk and k1.diagonal_shift == (0.0, 0.0, 0.0)
    k2 = it.regularize_equality_gradient(kkt=k1, coef=-1e-8, copy_kkt=False)
    k2 = it.regularize_hessian(kkt=k2, coef=1e-8, copy_kkt=False)
    assert k2.base is dk and k2.diagonal_shift == (1e-8, 1e-8, 1e-8)
    k3 = it.regularize_hessian(kkt=it.regularize_equality_gradient(kkt=k2, coef=-1e-7, copy_kkt=False), coef=1e-7,
                               copy_kkt=False)
    # the reference adds to the Hessian block and replaces the others (interface.py:590-619)
    assert k3.diagonal_shift == (1e-8 + 1e-7, 1e-7, 1e-7)


# ---- the producer and the loop on the CPU engines (numpy restatement of the kernels + the host interpreter) ---------------
def _cpu_device_loop(qps, fs, comm=None):
    from hostsim_engine import HostSimDeviceEngine
    from parapint_amd.algorithms.device_interior_point import ip_solve_device
    from parapint_amd.algorithms.interior_point import IPOptions, InteriorPointStatus
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
    it = DeviceStochasticQPInterface(qps, fs, comm=comm)
    options = IPOptions()
    options.linalg.solver = HipSchurComplementLinearSolver({i: None for i in it.local}, None, comm=comm or SerialComm(),
                                                           engine=HostSimDeviceEngine())
    hist = []
    status, _ = ip_solve_device(it, options, history=hist)
    assert status == InteriorPointStatus.optimal
    return it, hist


def _oracle_host_loop(qps, fs):
    """The restated reference loop over the ORACLE's solver classes (the reference algorithm on SuperLU)."""
    from oracle.schur_complement import SchurComplementLinearSolver as OracleSC
    from oracle.subsolvers import ScipyInterface as OracleScipy
    N = len(qps)
    return host_loop(qps, fs, OracleSC({i: OracleScipy(compute_inertia=True) for i in range(N)},
                                       OracleScipy(compute_inertia=True)))


def _same_run(rows, hist, hi, it, n_scenarios, same_solver=True, tol=None):
    """same_solver: both loops factorise with the same solver class and must agree iteration by iteration.  Over the
    oracle's classes (SuperLU, inertia from dense eigenvalues) the inertia-correction loop regularises an iterate or two
    that the LDL^T inertia accepts -- the runs part early and must arrive at the same point."""
    if same_solver:
        assert len(rows) == len(hist)                            # same number of iterations
        for r, h in zip(rows, hist):
            # the host loop prints 3 significant digits: primal / dual / complementarity infeasibility and the barrier
            # (below 1e-9 the measures are rounding noise of two different summation orders)
            for a, b in zip(r[2:6], (h[0], h[1], h[2], h[3])):
                assert abs(a - b) <= 6e-3 * max(abs(a), abs(b)) + 2e-9
    else:
        assert abs(len(rows) - len(hist)) <= max(6, len(rows) // 3)
        for a, b in zip(rows[0][2:6], hist[0][:4]):              # the same initial point and measures
            assert abs(a - b) <= 6e-3 * max(abs(a), abs(b)) + 2e-9
    # (two runs that stop at 1e-8 on different paths agree to the accuracy the stopping test gives the variables)
    if tol is None:
        tol = 1.0 if same_solver else 50.0
    zh = np.asarray(hi.get_primals().get_block(n_scenarios))
    assert np.abs(it.first_stage_solution() - zh).max() <= tol * 1e-7 * max(1.0, np.abs(zh).max())
    for ndx in (0, n_scenarios - 1):
        xh = hi.scenario_interface(ndx).get_primals()
        assert np.abs(it.scenario_primals(ndx) - xh).max() <= tol * 1e-6 * max(1.0, np.abs(xh).max())


def _interpreter_host_loop(qps, fs):
    """The restated reference loop over the product's solver class on the host interpreter (host containers)."""
    from hostsim_engine import HostSimEngine
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
    return host_loop(qps, fs, HipSchurComplementLinearSolver({i: None for i in range(len(qps))}, None, comm=SerialComm(),
                                                             engine=HostSimEngine()))


def test_device_loop_on_cpu_engines_matches_the_host_loops():
    qps, fs = random_stochastic_qp(6, seed=2)
    it, hist = _cpu_device_loop(qps, fs)
    hi, rows = _interpreter_host_loop(qps, fs)
    _same_run(rows, hist, hi, it, len(qps))
    hi, rows = _oracle_host_loop(qps, fs)
    _same_run(rows, hist, hi, it, len(qps), same_solver=False)


def test_device_loop_on_cpu_engines_against_the_oracle_loop_at_64_scenarios_tight_gate():
    qps, fs = random_stochastic_qp(64, seed=2)
    it, hist = _cpu_device_loop(qps, fs)
    hi, rows = _oracle_host_loop(qps, fs)
    _same_run(rows, hist, hi, it, 64, same_solver=False, tol=5.0)


def test_two_pattern_groups_on_cpu_engines():
    from ip_multirank_worker import mixed_scenarios
    qps, fs = mixed_scenarios()
    it, hist = _cpu_device_loop(qps, fs)
    assert len(it.pattern_groups) == 2 and len(it.states) == 2 and it.nsrc == max(pg.nsrc for pg in it.pattern_groups)
    hi, rows = _interpreter_host_loop(qps, fs)
    _same_run(rows, hist, hi, it, len(qps))
    hi, rows = _oracle_host_loop(qps, fs)
    _same_run(rows, hist, hi, it, len(qps), same_solver=False)


def test_stacked_initial_points_are_those_of_the_host_loop():
    """The processed initial point of all scenarios of a pattern group at once (rows = scenarios) against the host
    interface's per-scenario initialisation and the loop's process_init (interior_point.py:433-447, 761-799): bitwise,
    with starting points outside and on their bounds, one-sided and free variables."""
    from parapint_amd.algorithms import interior_point as host_ip
    from parapint_amd.interfaces.interface import _relaxed
    qps, fs = random_stochastic_qp(5, n=18, n_fs=3, n_eq=4, n_ineq=6, seed=11)
    rng = np.random.default_rng(3)
    for q in qps:
        q.x0 = rng.normal(size=q.n) * 3
        q.lb[rng.random(q.n) < 0.3] = -np.inf
        q.ub[rng.random(q.n) < 0.3] = np.inf
        on = np.flatnonzero(np.isfinite(q.lb))[:2]
        q.x0[on] = q.lb[on]                                        # exactly on a bound
    it = DeviceStochasticQPInterface(qps, fs, bounds_relaxation_factor=1e-8)
    pg = it.pattern_groups[0]
    st = pg.initial_points(qps, 1e-8)
    for b, q in enumerate(qps):
        lb, ub = _relaxed(q.lb, 1e-8, -1.0), _relaxed(q.ub, 1e-8, +1.0)
        ilb, iub = _relaxed(q.ineq_lb, 1e-8, -1.0), _relaxed(q.ineq_ub, 1e-8, +1.0)
        x, s = q.x0.copy(), np.asarray(q.A_ineq @ q.x0, dtype=np.double)
        zl, zu = np.ones(q.n), np.ones(q.n)
        zl[np.isneginf(q.lb)] = 0
        zu[np.isinf(q.ub)] = 0
        sl, su = np.zeros(s.size), np.zeros(s.size)
        host_ip.process_init(x, lb, ub); host_ip.process_init(s, ilb, iub)
        host_ip.process_init_duals_lb(zl, lb); host_ip.process_init_duals_ub(zu, ub)
        host_ip.process_init_duals_lb(sl, ilb); host_ip.process_init_duals_ub(su, iub)
        for k, v in dict(x=x, s=s, zl=zl, zu=zu, sl=sl, su=su, lb=lb, ub=ub, ilb=ilb, iub=iub).items():
            assert np.array_equal(st[k][b], v), (b, k)
    one = pg.initial_point(qps[2], 1e-8)
    assert all(np.array_equal(one[k], st[k][2]) for k in st)


def test_row_programs_reproduce_the_host_interface():
    """grad f + J^T y, the constraint residuals and the right-hand side of the numpy kernels (through the row programs)
    against the host interface's own evaluation at a random iterate."""
    from hostsim_ip_ops import HostSimIpOps
    from parapint_amd.interfaces.schur_complement.device_sc_ip_interface import _PatternGroup
    from parapint_amd.interfaces.schur_complement.sc_ip_interface import StochasticSchurComplementInteriorPointInterface
    from parapint_amd.sparse.block_containers import BlockVector
    rng = np.random.default_rng(9)
    qps, fs = random_stochastic_qp(3, seed=6)
    pg = _PatternGroup(qps[0], fs[0])
    n, mi, me, nfs, nb = pg.n, pg.mi, pg.me, pg.nfs, pg.nb
    prog, terms = pg.row_programs()
    B, bpad = 3, 64
    W = np.zeros((nb + 2 * n + 2 * mi, bpad))
    W[:, :B] = rng.uniform(0.5, 1.5, size=(W.shape[0], B))
    bounds = np.zeros((2 * n + 2 * mi, bpad))
    data = np.zeros((n + me, bpad))
    src = np.zeros((pg.nsrc, bpad))
    host = StochasticSchurComplementInteriorPointInterface(qps, fs)
    host.set_bounds_relaxation_factor(1e-8)
    z = rng.normal(size=nfs)
    mu = 0.05
    for b, q in enumerate(qps):
        nlp = host.scenario_interface(b)
        bounds[0:n, b], bounds[n:2 * n, b] = nlp.primals_lb(), nlp.primals_ub()
        bounds[2 * n:2 * n + mi, b], bounds[2 * n + mi:, b] = nlp.ineq_lb(), nlp.ineq_ub()
        lo, hi = bounds[0:n, b], bounds[n:2 * n, b]
        W[0:n, b] = np.where(np.isfinite(lo), lo + 0.7, np.where(np.isfinite(hi), hi - 0.7, W[0:n, b]))
        data[0:n, b], data[n:, b] = q.c, q.b_eq
        src[pg.off[0]:pg.off[1], b], src[pg.off[1]:pg.off[2], b], src[pg.off[2]:pg.off[3], b] = q.H.data, q.A_eq.data, q.A_ineq.data
        nlp.set_primals(W[0:n, b]); nlp.set_slacks(W[n:n + mi, b])
        nlp.set_duals_eq(W[n + mi:n + mi + me, b]); nlp.set_duals_ineq(W[n + mi + me:n + 2 * mi + me, b])
        host._duals_link[b] = W[n + 2 * mi + me:nb, b].copy()
    host._primals_coupling = z.copy()
    host.set_barrier_parameter(mu)
    d = dict(n=n, mi=mi, me=me, nfs=nfs, batch=B, bpad=bpad, src_dp=int(pg.off[3]), src_ds=int(pg.off[4]), W=W, bounds=bounds,
             data=data, src=src, G=np.zeros((n, bpad)), rhs=np.zeros((nb, bpad)), prog=prog, terms=terms)
    ops = HostSimIpOps()
    hd = ops.prepare([d])
    v, rc = np.zeros(8 + nfs), np.zeros(nfs)
    ops.take_step(hd, None, 1, False, mu, z, None)
    ops.residuals(hd, z, v)
    ops.publish(v, None, 1, nfs, 0, rc)
    ops.rhs(hd, mu)
    rhs = host.evaluate_primal_dual_kkt_rhs()
    for b in range(B):
        ref = rhs.get_block(b).flatten()
        got = hd.descs[0]['rhs'][:, b]
        assert np.allclose(got, ref, rtol=1e-12, atol=1e-12), np.abs(got - ref).max()
    assert np.allclose(rc, rhs.get_block(B), rtol=1e-12, atol=1e-13)
    obj = sum(host.scenario_interface(b).evaluate_objective() - qps[b].c0 for b in range(B))
    assert abs(ops.wait()[6] - obj) <= 1e-12 * max(1.0, abs(obj))


@pytest.mark.gpu
def test_two_rank_device_loop_on_the_device():
    """The same with the real kernels: two ranks share the device(s), the collectives (S, r_s, the two all-gathers of the
    step) go through gloo -- the rehearsal of the N > 1 path a one-GPU box allows."""
    _two_rank_run('--gpu')


def test_two_rank_device_loop():
    """world_size 2 over gloo: rank-distributed scenarios of two sparsity patterns; the iterates, the iteration count and
    every measure equal those of the one-rank run, and both ranks publish identical scalars."""
    _two_rank_run()


def _two_rank_run(*args):
    import os
    import socket
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(here, 'ip_multirank_worker.py')] + list(args)
    env = dict(os.environ)
    env['OMP_NUM_THREADS'] = '1'
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = out.stdout.decode()
    assert out.returncode == 0, text[-4000:]
    assert 'rank 0 ok' in text and 'rank 1 ok' in text


class _Capture(logging.Handler):
    def __init__(self):
        logging.Handler.__init__(self)
        self.rows = []

    def emit(self, record):
        m = re.match(r'^(\d+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)', record.getMessage())
        if m:
            self.rows.append([float(v) for v in m.groups()])


def host_loop(qps, fs, solver):
    from parapint_amd.algorithms.interior_point import IPOptions, InteriorPointStatus, ip_solve
    from parapint_amd.interfaces.schur_complement.sc_ip_interface import StochasticSchurComplementInteriorPointInterface
    interface = StochasticSchurComplementInteriorPointInterface(qps, fs)
    options = IPOptions()
    options.linalg.solver = solver
    log = logging.getLogger('parapint_amd.algorithms.interior_point')
    cap = _Capture()
    old = log.level
    log.addHandler(cap)
    log.setLevel(logging.INFO)
    try:
        status = ip_solve(interface=interface, options=options)
    finally:
        log.removeHandler(cap)
        log.setLevel(old)
    assert status == InteriorPointStatus.optimal
    return interface, cap.rows


def device_loop(qps, fs):
    from parapint_amd.algorithms.device_interior_point import ip_solve_device
    from parapint_amd.algorithms.interior_point import IPOptions, InteriorPointStatus
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
    it = DeviceStochasticQPInterface(qps, fs)
    options = IPOptions()
    options.linalg.solver = HipSchurComplementLinearSolver({i: None for i in range(len(qps))}, None, comm=SerialComm())
    hist = []
    status, iters = ip_solve_device(it, options, history=hist)
    assert status == InteriorPointStatus.optimal
    return it, hist, options.linalg.solver


@pytest.mark.gpu
def test_farmer_through_the_device_loop():
    qps, fs = farmer_qps(extra=1)
    it, hist, solver = device_loop(qps, fs)
    assert np.abs(it.first_stage_solution() - np.array([170.0, 80.0, 250.0])).max() < 5e-6
    for ndx in range(len(qps)):
        assert np.abs(it.scenario_primals(ndx)[:3] - np.array([170.0, 80.0, 250.0])).max() < 5e-6
    # (with the per-entry zero-pivot test of round 3 the farmer's KKT matrices are never reported singular: no retries)


@pytest.mark.gpu
@pytest.mark.parametrize('n_scenarios', [8, 256])
def test_stochastic_qp_device_loop_matches_host_loop(n_scenarios):
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
    qps, fs = random_stochastic_qp(n_scenarios, seed=2)
    it, hist, _ = device_loop(qps, fs)
    host_solver = HipSchurComplementLinearSolver({i: None for i in range(n_scenarios)}, None, comm=SerialComm())
    hi, rows = host_loop(qps, fs, host_solver)
    _same_run(rows, hist, hi, it, n_scenarios)


@pytest.mark.gpu
def test_device_loop_against_the_host_loop_over_the_oracle_solver():
    """8 scenarios: the device loop against the restated reference loop over the ORACLE's solver classes (the reference
    algorithm on SuperLU, inertia from dense eigenvalues) -- same initial measures, same solution."""
    qps, fs = random_stochastic_qp(8, seed=2)
    it, hist, _ = device_loop(qps, fs)
    hi, rows = _oracle_host_loop(qps, fs)
    _same_run(rows, hist, hi, it, 8, same_solver=False)


@pytest.mark.gpu
def test_device_loop_against_the_oracle_loop_at_64_scenarios_tight_gate():
    """64 scenarios: the device loop (every back-solve checked) against the restated reference loop over the oracle's
    solver classes -- the first-stage solution within 5e-7, the scenario primals within 5e-6 (round 5: 50 x that, 8 scenarios)."""
    qps, fs = random_stochastic_qp(64, seed=2)
    it, hist, _ = device_loop(qps, fs)
    hi, rows = _oracle_host_loop(qps, fs)
    _same_run(rows, hist, hi, it, 64, same_solver=False, tol=5.0)


@pytest.mark.gpu
def test_rank_deficient_constraints_go_through_the_retries_from_resident_values():
    """Every scenario states one equality constraint twice: the KKT matrix is singular at every iterate, the
    factorisation must say so (one zero eigenvalue per scenario) and the inertia-correction loop regularises -- on the
    device through `do_numeric_factorization(matrix + diagonal)` from the resident values, iteration by
    iteration like the host loop."""
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
    n_scenarios = 16
    qps, fs = random_stochastic_qp(n_scenarios, seed=4, duplicate_eq_row=True)
    it, hist, solver = device_loop(qps, fs)
    # (a retry in nearly every iteration: the duplicated row's pivot is exactly zero up to rounding, and now and then the
    # rounding leaves it above the bound)
    assert solver.diagonal_shift_refactorizations >= (len(hist) - 1) // 2
    host_solver = HipSchurComplementLinearSolver({i: None for i in range(n_scenarios)}, None, comm=SerialComm())
    hi, rows = host_loop(qps, fs, host_solver)
    # (a regularised singular system is ill-conditioned: the two loops, whose sums run in different orders, may part by an
    # iteration near the end; they must arrive at the same point)
    assert abs(len(rows) - len(hist)) <= 2
    zh = np.asarray(hi.get_primals().get_block(n_scenarios))
    assert np.abs(it.first_stage_solution() - zh).max() <= 1e-5 * max(1.0, np.abs(zh).max())


# ---- the kernels of the interior-point step against their numpy restatement (tests/hostsim_ip_ops.py) ------------------
def _step_problem(seed, shapes):
    """Random iterates, bounds (some infinite), data, sources and steps for pattern groups of the given
    (instances, n, n_fs, n_eq, n_ineq); host arrays in the kernels' [row][instance] layout."""
    from parapint_amd.interfaces.schur_complement.device_sc_ip_interface import _PatternGroup
    rng = np.random.default_rng(seed)
    descs = []
    nfs = shapes[0][2]
    for gi, (B, n, _, me, mi) in enumerate(shapes):
        qps, fs = random_stochastic_qp(1, n=n, n_fs=nfs, n_eq=me, n_ineq=mi, seed=seed + gi)
        pg = _PatternGroup(qps[0], fs[0])
        prog, terms = pg.row_programs()
        bpad = -(-B // 64) * 64
        nb = pg.nb
        W = rng.uniform(0.5, 2.0, size=(nb + 2 * n + 2 * mi, bpad))
        W[n + mi:nb] = rng.normal(size=(nb - n - mi, bpad))                      # constraint duals: any sign
        lo = W[:n + mi] - rng.uniform(0.1, 1.0, size=(n + mi, bpad))
        hi = W[:n + mi] + rng.uniform(0.1, 1.0, size=(n + mi, bpad))
        lo[rng.random(lo.shape) < 0.3] = -np.inf
        hi[rng.random(hi.shape) < 0.3] = np.inf
        bounds = np.concatenate([lo[:n], hi[:n], lo[n:], hi[n:]])
        zl = np.concatenate([W[nb:nb + n], W[nb + 2 * n:nb + 2 * n + mi]])
        zu = np.concatenate([W[nb + n:nb + 2 * n], W[nb + 2 * n + mi:]])
        zl[~np.isfinite(lo)] = 0.0
        zu[~np.isfinite(hi)] = 0.0
        W[nb:nb + n], W[nb + 2 * n:nb + 2 * n + mi] = zl[:n], zl[n:]
        W[nb + n:nb + 2 * n], W[nb + 2 * n + mi:] = zu[:n], zu[n:]
        src = np.zeros((pg.nsrc, bpad))
        src[:pg.off[3]] = rng.normal(size=(pg.off[3], bpad))
        descs.append(dict(n=n, mi=mi, me=me, nfs=nfs, batch=B, bpad=bpad, src_dp=int(pg.off[3]), src_ds=int(pg.off[4]),
                          W=W, bounds=bounds, data=rng.normal(size=(n + me, bpad)), src=src, G=np.zeros((n, bpad)),
                          rhs=np.zeros((nb, bpad)), prog=prog, terms=terms, delta=rng.normal(size=(nb, bpad))))
    return descs, rng.normal(size=nfs), rng.normal(size=nfs)


def _run_step_sequence(ops, descs, z, dz, mu, tau, nranks=1, ncoup=None, dual_from=0):
    """take_step (no step) -> residuals -> publish -> rhs -> step lengths -> take_step -> residuals -> publish.
    ncoup / dual_from: rows of the coupling block and where its variables start (mapped groups: 2 ncz, ncz)."""
    dev = []
    for d in descs:
        dd = {k: (ops.from_host(v) if isinstance(v, np.ndarray) else v) for k, v in d.items() if k != 'delta'}
        dev.append(dd)
    hd = ops.prepare(dev)
    nfs = descs[0]['nfs'] if ncoup is None else ncoup
    zt, dzt = ops.from_host(z.copy()), ops.from_host(dz.copy())
    alpha, v = ops.zeros((2,)), ops.zeros((8 + nfs,))
    rc = ops.zeros((max(nfs, 1),))
    out = {}
    ops.take_step(hd, None, 1, False, mu, zt, None)
    ops.residuals(hd, zt, v)
    ops.publish(v, None, 1, nfs, dual_from, rc)
    out['mail0'] = ops.wait()
    out['v0'] = ops.to_host(v).copy()
    ops.rhs(hd, mu)
    out['rhs0'] = [ops.to_host(d['rhs']).copy() for d in dev]
    out['src0'] = [ops.to_host(d['src']).copy() for d in dev]
    for gi, d in enumerate(descs):
        ops.set_delta(hd, gi, ops.from_host(d['delta']))
    ops.step_lengths(hd, tau, mu, alpha)
    out['alpha'] = ops.to_host(alpha).copy()
    ops.take_step(hd, alpha, 1, False, mu, zt, dzt)
    ops.residuals(hd, zt, v)
    ops.publish(v, alpha, 1, nfs, dual_from, rc)
    out['mail1'] = ops.wait()
    out['W1'] = [ops.to_host(d['W']).copy() for d in dev]
    out['G1'] = [ops.to_host(d['G']).copy() for d in dev]
    ops.rhs(hd, mu)
    out['rhs1'] = [ops.to_host(d['rhs']).copy() for d in dev]
    out['src1'] = [ops.to_host(d['src']).copy() for d in dev]
    out['z1'] = ops.to_host(zt).copy()
    out['rc1'] = ops.to_host(rc).copy()
    return out


@pytest.mark.gpu
def test_step_kernels_with_a_model_supplied_objective_row():
    """obj_row >= 0 (nonlinear models: the objective value of every instance is a data row the caller's model wrote) on
    groups WITHOUT a coupling map: the objective of the mailbox is the sum of that row over the instances of all groups,
    everything else as for a QP."""
    from hostsim_ip_ops import HostSimIpOps
    from parapint_amd.linalg.hip_schur_complement import HipEngine
    shapes = [(70, 33, 5, 9, 0), (3, 12, 5, 4, 7)]
    descs, z, dz = _step_problem(19, shapes)
    rng = np.random.default_rng(2)
    for d in descs:
        extra = rng.normal(size=(1, d['data'].shape[1]))
        d['data'] = np.concatenate([d['data'], extra])
        d['obj_row'] = d['data'].shape[0] - 1
    want = sum(float(d['data'][d['obj_row'], :d['batch']].sum()) for d in descs)
    mu, tau = 0.1, 0.9
    ref = _run_step_sequence(HostSimIpOps(), [dict(d, **{k: v.copy() for k, v in d.items() if isinstance(v, np.ndarray)})
                                              for d in descs], z, dz, mu, tau)
    got = _run_step_sequence(HipEngine().ip_ops(), descs, z, dz, mu, tau)
    for key in ('mail0', 'mail1'):
        assert abs(got[key][6] - want) <= 1e-12 * max(1.0, abs(want)) and abs(ref[key][6] - want) <= 1e-12 * max(1.0, abs(want))
        assert np.array_equal(got[key][[0, 2, 3, 7, 8]], ref[key][[0, 2, 3, 7, 8]])
    for key in ('W1', 'G1', 'rhs1', 'src1'):
        for gi, (a, b) in enumerate(zip(got[key], ref[key])):
            assert np.array_equal(a[:, :shapes[gi][0]], b[:, :shapes[gi][0]])


@pytest.mark.gpu
@pytest.mark.parametrize('shapes', [[(5, 24, 4, 6, 8)], [(70, 33, 5, 9, 0), (3, 12, 5, 4, 7)], [(130, 40, 3, 0, 5)]])
def test_step_kernels_match_their_numpy_restatement(shapes):
    """Elementwise results (iterate, barrier diagonals, right-hand side, grad f + J^T y), step lengths and max-norms bit
    for bit; sums (duals, objective, coupling block) to rounding of a different summation order."""
    from hostsim_ip_ops import HostSimIpOps
    from parapint_amd.linalg.hip_schur_complement import HipEngine
    descs, z, dz = _step_problem(11, shapes)
    mu, tau = 0.1, 0.9
    ref = _run_step_sequence(HostSimIpOps(), [dict(d, **{k: v.copy() for k, v in d.items() if isinstance(v, np.ndarray)})
                                              for d in descs], z, dz, mu, tau)
    got = _run_step_sequence(HipEngine().ip_ops(), descs, z, dz, mu, tau)
    assert np.array_equal(got['alpha'], ref['alpha']) and 0.0 < ref['alpha'].min() < 1.0
    for key in ('rhs0', 'src0', 'W1', 'G1', 'rhs1', 'src1'):
        for gi, (a, b) in enumerate(zip(got[key], ref[key])):
            B = shapes[gi][0]
            assert np.array_equal(a[:, :B], b[:, :B]), (key, gi, np.abs(a[:, :B] - b[:, :B]).max())
    assert np.array_equal(got['z1'], ref['z1'])
    for key in ('mail0', 'mail1'):
        a, b = got[key], ref[key]
        assert np.array_equal(a[[0, 2, 3, 7, 8]], b[[0, 2, 3, 7, 8]]), (key, a, b)            # max / min: exact
        # sums, and the dual infeasibility (its coupling part is |sum over the instances of y_link|)
        assert np.allclose(a[[1, 4, 5, 6]], b[[1, 4, 5, 6]], rtol=1e-13, atol=0.0), (key, a, b)
    assert np.allclose(got['rc1'], ref['rc1'], rtol=1e-13, atol=1e-13)
    assert np.allclose(got['v0'], ref['v0'], rtol=1e-13, atol=1e-13)


@pytest.mark.gpu
def test_initial_point_processing_on_the_device_matches_the_host_functions():
    """relax_bounds + process_initial_point over [row][lane] tensors against the host loop's functions on the same
    arrays (bit for bit): points outside, on and inside their bounds, one-sided, free and padded lanes; crossed and
    equal bounds raise the host loop's ValueErrors."""
    from hostsim_ip_ops import HostSimIpOps
    from parapint_amd.linalg.hip_schur_complement import HipEngine
    rng = np.random.default_rng(5)
    n, mi, bpad = 37, 6, 128
    nb = n + 2 * mi + 3 + 4
    lo = rng.normal(size=(n + mi, bpad))
    hi = lo + rng.uniform(0.1, 4.0, size=lo.shape)
    lo[rng.random(lo.shape) < 0.3] = -np.inf
    hi[rng.random(hi.shape) < 0.3] = np.inf
    x = rng.normal(size=lo.shape) * 3
    on = np.isfinite(lo) & (rng.random(lo.shape) < 0.1)
    x[on] = lo[on]
    lo[:, 100:], hi[:, 100:] = -np.inf, np.inf                      # padded lanes
    W = np.zeros((nb + 2 * n + 2 * mi, bpad))
    W[:n + mi] = x
    bounds = np.concatenate([lo[:n], hi[:n], lo[n:], hi[n:]])
    ref_ops, dev_ops = HostSimIpOps(), HipEngine().ip_ops()
    Wr, br = W.copy(), bounds.copy()
    ref_ops.relax_bounds(br, n, mi, 1e-8)
    ref_ops.process_initial_point(Wr, br, n, mi, nb)
    Wd, bd = dev_ops.from_host(W), dev_ops.from_host(bounds)
    dev_ops.relax_bounds(bd, n, mi, 1e-8)
    dev_ops.process_initial_point(Wd, bd, n, mi, nb)
    assert np.array_equal(dev_ops.to_host(bd), br) and np.array_equal(dev_ops.to_host(Wd), Wr)
    assert (Wr[:n] > br[:n]).all() and (Wr[:n] < br[n:2 * n]).all() and not np.array_equal(Wr[:n + mi], x)
    assert set(np.unique(Wr[nb:])) == {0.0, 1.0} and (Wr[nb:, 100:] == 0).all()
    for bad, msg in ((-1.0, 'larger than upper'), (0.0, 'equal lower and upper')):
        b2 = bounds.copy()
        b2[3, 7], b2[n + 3, 7] = 1.0, 1.0 + bad
        for ops in (ref_ops, dev_ops):
            with pytest.raises(ValueError, match=msg):
                ops.process_initial_point(ops.from_host(W), ops.from_host(b2), n, mi, nb)
