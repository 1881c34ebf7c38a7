"""A NONLINEAR time-staged problem through the reference's interfaces: optimal control of the viscous Burgers equation
(parapint/examples/burgers.py:53-176 restated by hand as an NLP object, parapint_amd/examples/burgers.py), i.e.
``InteriorPointInterface`` over the NLP protocol (interfaces/interface.py:251-679) under
``MPIDynamicSchurComplementInteriorPointInterface`` -- Hessian and Jacobian values change at every iterate."""
import numpy as np
import pytest

from parapint_amd.examples import burgers as bg


def _oracle_solver(blocks):
    from oracle.schur_complement import SchurComplementLinearSolver as OracleSC
    from oracle.subsolvers import ScipyInterface as OracleScipy
    return OracleSC({i: OracleScipy(compute_inertia=True) for i in blocks}, OracleScipy(compute_inertia=True))


def test_burgers_nlp_derivatives_against_finite_differences():
    for init in (True, False):
        nlp = bg.BurgersNLP(8, 3, 0.0, 0.25, init)
        rng = np.random.default_rng(0)
        x, lam = rng.normal(size=nlp.n_primals()), rng.normal(size=nlp.n_eq_constraints())
        nlp.set_primals(x)
        nlp.set_duals_eq(lam)
        g, J = nlp.evaluate_grad_objective(), nlp.evaluate_jacobian_eq().toarray()
        H = nlp.evaluate_hessian_lag().toarray()
        assert np.array_equal(H, np.tril(H)) and nlp.nnz_hessian_lag() == nlp.evaluate_hessian_lag().nnz
        H = H + np.tril(H, -1).T

        def at(z, fn):
            nlp.set_primals(z)
            return fn()
        eps, E = 1e-6, np.eye(nlp.n_primals())
        gn = np.array([(at(x + eps * e, nlp.evaluate_objective) - at(x - eps * e, nlp.evaluate_objective)) / (2 * eps) for e in E])
        Jn = np.array([(at(x + eps * e, nlp.evaluate_eq_constraints) - at(x - eps * e, nlp.evaluate_eq_constraints)) / (2 * eps)
                       for e in E]).T
        gl = lambda: nlp.evaluate_grad_objective() + nlp.evaluate_jacobian_eq().T @ lam
        Hn = np.array([(at(x + eps * e, gl) - at(x - eps * e, gl)) / (2 * eps) for e in E]).T
        assert np.abs(g - gn).max() <= 1e-8 and np.abs(J - Jn).max() <= 1e-7 and np.abs(H - Hn).max() <= 1e-7


def _monolithic(nfe_x, nfe_t):
    """The whole horizon as ONE NLP (no time blocks, plain trapezoid objective) through the same loop."""
    from parapint_amd.algorithms.interior_point import IPOptions, InteriorPointStatus, ip_solve
    from parapint_amd.interfaces.schur_complement.sc_ip_interface import StochasticSchurComplementInteriorPointInterface
    nlp = bg.BurgersNLP(nfe_x, nfe_t, 0.0, 1.0, True, start_term=False)
    mono = StochasticSchurComplementInteriorPointInterface([nlp], [[0]])
    opt = IPOptions()
    opt.linalg.solver = _oracle_solver([0])
    assert ip_solve(mono, opt) == InteriorPointStatus.optimal
    return mono, nlp


def _same_as_monolithic(it, mono, nlp, T, nfe_t, tol):
    nb, m = nfe_t // T, nlp.m
    x = mono.get_primals().get_block(0)
    Y, U = x[:(nfe_t + 1) * m].reshape(nfe_t + 1, m), x[(nfe_t + 1) * m:].reshape(nfe_t + 1, m)
    assert abs(it.evaluate_objective() - mono.evaluate_objective()) <= tol
    for t in it.local_block_indices:
        xb = np.asarray(it.get_primals().get_block(t))
        y, u = xb[:(nb + 1) * m].reshape(nb + 1, m), xb[(nb + 1) * m:].reshape(nb + 1, m)
        assert np.abs(y - Y[t * nb:(t + 1) * nb + 1]).max() <= tol
        assert np.abs(u[:nb] - U[t * nb:(t + 1) * nb]).max() <= tol
        assert np.abs(u[nb]).max() <= tol             # the copy of the end node's control drives nothing
    z = np.asarray(it.get_primals().get_block(T))
    for t in range(T - 1):
        assert np.abs(z[m * t:m * (t + 1)] - Y[(t + 1) * nb]).max() <= tol
    assert np.abs(Y).max() > 0.5 and np.abs(U).max() > 0.05       # (a non-trivial state and control)


def test_burgers_time_blocks_match_the_monolithic_problem():
    T, nfe_x, nfe_t = 4, 8, 12
    it = bg.main(_oracle_solver(range(T)), nfe_x=nfe_x, nfe_t=nfe_t, nblocks=T)
    mono, nlp = _monolithic(nfe_x, nfe_t)
    _same_as_monolithic(it, mono, nlp, T, nfe_t, 1e-7)


def test_burgers_over_the_product_solver_on_the_cpu_engine():
    from hostsim_engine import HostSimEngine
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
    T, nfe_x, nfe_t = 3, 7, 9
    solver = HipSchurComplementLinearSolver({t: None for t in range(T)}, None, comm=SerialComm(), engine=HostSimEngine())
    it = bg.main(solver, nfe_x=nfe_x, nfe_t=nfe_t, nblocks=T)
    mono, nlp = _monolithic(nfe_x, nfe_t)
    _same_as_monolithic(it, mono, nlp, T, nfe_t, 1e-6)      # (two runs that stop at 1e-8 on different paths)


def _trajectory(it, T, nfe_t, m):
    """States y over the whole horizon from the time blocks (the shared time nodes once)."""
    nb = nfe_t // T
    rows = []
    for t in range(T):
        xb = np.asarray(it.get_primals().get_block(t))
        y = xb[:(nb + 1) * m].reshape(nb + 1, m)
        rows.append(y if t == 0 else y[1:])
    return np.concatenate(rows)


@pytest.mark.gpu
def test_burgers_over_the_hip_solver():
    """New Hessian and Jacobian values from the host at every iteration (the SURVEY 8(d) boundary with values that change
    everywhere): 6 time blocks against the same loop over the oracle's solver classes; then 16 and 8 time blocks over the
    same 128 steps x 39 grid points (coupling blocks 1170 -- block-tridiagonal S -- and 546): the optimal trajectory does
    not depend on where the horizon is cut."""
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver

    def hip(T):
        return HipSchurComplementLinearSolver({t: None for t in range(T)}, None, comm=SerialComm())
    T, nfe_x, nfe_t = 6, 16, 24
    it = bg.main(hip(T), nfe_x=nfe_x, nfe_t=nfe_t, nblocks=T)
    ref = bg.main(_oracle_solver(range(T)), nfe_x=nfe_x, nfe_t=nfe_t, nblocks=T)
    assert abs(it.evaluate_objective() - ref.evaluate_objective()) <= 1e-8
    # (the controls are weakly determined -- their curvature is omega dx dt ~ 5e-5 --, so two runs that stop at a scaled
    # dual infeasibility of 1e-8 agree in them to ~1e-4; the states follow the controls through the dynamics)
    for t in range(T + 1):
        assert np.abs(np.asarray(it.get_primals().get_block(t)) - np.asarray(ref.get_primals().get_block(t))).max() <= 2e-4
    nfe_x, nfe_t = 40, 128
    s16, s8 = hip(16), hip(8)
    a = bg.main(s16, nfe_x=nfe_x, nfe_t=nfe_t, nblocks=16)
    b = bg.main(s8, nfe_x=nfe_x, nfe_t=nfe_t, nblocks=8)
    assert s16._btd is not None and s8._btd is None
    assert abs(a.evaluate_objective() - b.evaluate_objective()) <= 1e-8
    ya, yb = _trajectory(a, 16, nfe_t, nfe_x - 1), _trajectory(b, 8, nfe_t, nfe_x - 1)
    assert ya.shape == yb.shape == (nfe_t + 1, nfe_x - 1) and np.abs(ya - yb).max() <= 2e-4 and np.abs(ya).max() > 0.5


@pytest.mark.gpu
def test_burgers_at_the_dimensions_of_baseline_configuration_4():
    """BASELINE.json configs[3] to the letter: the Burgers discretisation with 512 time blocks x 4018 variables (nfe_x = 50,
    40 time steps per block), 49 states between the blocks, coupling block 50 078 with block-tridiagonal S, through
    ``ip_solve`` with new Hessian / Jacobian values from the host at every iteration.  The loop's own convergence test (primal,
    dual and complementarity infeasibility <= 1e-8 of the WHOLE problem, evaluated by the interface) is the check."""
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
    T = 512
    solver = HipSchurComplementLinearSolver({t: None for t in range(T)}, None, comm=SerialComm(), result_buffers=2)
    it = bg.main(solver, nfe_x=50, nfe_t=T * 40, nblocks=T)
    assert solver._btd is not None and solver._nc == 2 * 49 * 511
    assert it.scenario_interface(1).n_primals() == 4018
    y = _trajectory(it, T, T * 40, 49)
    assert y.shape == (T * 40 + 1, 49) and np.isfinite(y).all() and 0.9 <= np.abs(y).max() <= 1.1
    assert 0.0 < it.evaluate_objective() < 0.1


# ---- device-resident iterates for the nonlinear problem (DeviceDynamicNLPInterface + BurgersDeviceModel) -----------------
def _device_nlp_loop(T, nfe_x, nfe_t, engine, comm=None):
    from parapint_amd.algorithms.device_interior_point import ip_solve_device
    from parapint_amd.algorithms.interior_point import IPOptions, InteriorPointStatus
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
    it = bg.device_interface(nfe_x, nfe_t, T, comm=comm)
    opt = IPOptions()
    opt.linalg.solver = HipSchurComplementLinearSolver({t: None for t in it.local}, None, comm=comm or SerialComm(), engine=engine,
                                                       result_buffers=0 if engine is not None else 2)
    hist, stats = [], {}
    status, iters = ip_solve_device(it, opt, history=hist, stats=stats)
    assert status == InteriorPointStatus.optimal
    return it, hist, stats


def _host_rows(T, nfe_x, nfe_t, solver):
    import logging
    import re
    rows = []

    class Cap(logging.Handler):
        def emit(self, record):
            m = re.match(r'^(\d+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)', record.getMessage())
            if m:
                rows.append([float(v) for v in m.groups()])
    log = logging.getLogger('parapint_amd.algorithms.interior_point')
    cap, old = Cap(), log.level
    log.addHandler(cap)
    log.setLevel(logging.INFO)
    try:
        it = bg.main(solver, nfe_x=nfe_x, nfe_t=nfe_t, nblocks=T)
    finally:
        log.removeHandler(cap)
        log.setLevel(old)
    return it, rows


def test_device_model_matches_the_nlp_object():
    """BurgersDeviceModel on [row][lane] arrays against BurgersNLP lane by lane: grad f, -c(x), objective, Jacobian and
    Hessian values in the NLP's entry order (bit for bit where the operation order is the same, else to rounding)."""
    rng = np.random.default_rng(4)
    for init in (False, True):
        nlps = [bg.BurgersNLP(9, 4, 0.25, 0.5, init) for _ in range(3)]
        q = nlps[0]
        n, me, bp = q.n_primals(), q.n_eq_constraints(), 4
        X, Lm = rng.normal(size=(n, bp)), rng.normal(size=(me, bp))
        nh, nj = q.nnz_hessian_lag(), q.nnz_jacobian_eq()
        W = np.zeros((n + me, bp))
        W[:n], W[n:] = X, Lm
        src, data = np.zeros((nh + nj, bp)), np.zeros((n + me + 1, bp))
        # (constants the producer writes at set-up: the Hessian diagonal and the Jacobian of the initial conditions)
        for b in range(bp):
            o = nlps[b % 3]
            o.set_primals(X[:, b]); o.set_duals_eq(Lm[:, b])
            src[:n, b] = o.evaluate_hessian_lag().data[:n]
            src[nh:, b] = o.evaluate_jacobian_eq().data
        src[n:nh] = 7.0
        src[nh:nh + q.nt * (5 * q.m - 2)] = 7.0
        bg.BurgersDeviceModel([nlps[b % 3] for b in range(bp)], bp).evaluate(W, src, data, dict(n=n, me=me, y_eq=n, hess=0, jac=nh,
                                                                                            obj_row=n + me))
        for b in range(bp):
            o = nlps[b % 3]
            o.set_primals(X[:, b]); o.set_duals_eq(Lm[:, b])
            assert np.allclose(data[:n, b], o.evaluate_grad_objective(), rtol=1e-14, atol=0)
            assert np.allclose(data[n:n + me, b], -o.evaluate_eq_constraints(), rtol=1e-12, atol=1e-13)
            assert abs(data[n + me, b] - o.evaluate_objective()) <= 1e-12 * max(1.0, abs(o.evaluate_objective()))
            assert np.allclose(src[:nh, b], o.evaluate_hessian_lag().data, rtol=1e-13, atol=0)
            assert np.allclose(src[nh:, b], o.evaluate_jacobian_eq().data, rtol=1e-13, atol=0)


def test_device_nonlinear_loop_on_cpu_engines_matches_the_host_loop():
    """The nonlinear problem with device-resident iterates (numpy engines): the iterations of the host loop over the NLP
    objects, measure for measure, and the same point."""
    from hostsim_engine import HostSimDeviceEngine, HostSimEngine
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
    T, nfe_x, nfe_t = 4, 8, 12
    it, hist, _ = _device_nlp_loop(T, nfe_x, nfe_t, HostSimDeviceEngine())
    host, rows = _host_rows(T, nfe_x, nfe_t, HipSchurComplementLinearSolver({t: None for t in range(T)}, None, comm=SerialComm(),
                                                                           engine=HostSimEngine()))
    assert len(rows) == len(hist)
    for r, h in zip(rows, hist):
        for a, b in zip(r[2:6], h[:4]):
            assert abs(a - b) <= 6e-3 * max(abs(a), abs(b)) + 2e-9, (r, h)
    assert abs(it.evaluate_objective() - host.evaluate_objective()) <= 1e-12
    for t in range(T):
        assert np.abs(it.scenario_primals(t) - np.asarray(host.get_primals().get_block(t))).max() <= 1e-10
    assert np.abs(it.coupling_states() - np.asarray(host.get_primals().get_block(T))).max() <= 1e-10


@pytest.mark.gpu
def test_device_nonlinear_loop_on_the_device():
    """The same on the HIP kernels (the model's functions are torch operations on the resident tensors -- the model is the
    caller's code, as Pyomo's is for the reference; the producer's own steps dispatch none): 16 time blocks x 8 steps x 39 grid
    points against the numpy engines, then BASELINE configs[3] to the letter against the host producer's optimum."""
    from hostsim_engine import HostSimDeviceEngine
    T, nfe_x, nfe_t = 16, 40, 128
    it, hist, stats = _device_nlp_loop(T, nfe_x, nfe_t, None)
    ref, ref_hist, _ = _device_nlp_loop(T, nfe_x, nfe_t, HostSimDeviceEngine())
    assert len(hist) == len(ref_hist)
    for a, b in zip(hist, ref_hist):
        assert np.allclose(a[:4], b[:4], rtol=1e-6, atol=2e-9), (a, b)
    for t in range(T):
        assert np.abs(it.scenario_primals(t) - ref.scenario_primals(t)).max() <= 1e-7
    T = 512
    it, hist, stats = _device_nlp_loop(T, 50, T * 40, None)
    assert len(hist) <= 8 and max(hist[-1][:3]) <= 1e-8 and it.pattern_groups[1].n == 4018 and 2 * it.ncz == 50078
    assert 0.0 < it.evaluate_objective() < 0.1


def test_device_nonlinear_loop_on_two_ranks():
    """Time blocks of the nonlinear problem dealt over two gloo ranks (numpy engines): the iterations and the point of the
    one-rank run."""
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
           '--master-port', str(port), os.path.join(here, 'dynamic_multirank_worker.py'), '--nonlinear']
    env = dict(os.environ)
    env['OMP_NUM_THREADS'] = '1'
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = out.stdout.decode()
    assert out.returncode == 0, text[-4000:]
    assert 'rank 0 ok' in text and 'rank 1 ok' in text


@pytest.mark.gpu
def test_hip_model_kernel_matches_the_array_form():
    """csrc/example_burgers.hip against the array form of the model: bit for bit against its numpy evaluation (the same
    operations in the same order; the objective sums in another order), to rounding against its torch evaluation on the same
    device tensors (torch divides by a scalar through its reciprocal)."""
    import torch
    rng = np.random.default_rng(9)
    for init in (False, True):
        bp = 128
        nlps = [bg.BurgersNLP(12, 5, 0.1 * b, 0.1 * (b + 1), init) for b in range(bp)]
        q = nlps[0]
        n, me = q.n_primals(), q.n_eq_constraints()
        nh, nj = q.nnz_hessian_lag(), q.nnz_jacobian_eq()
        Wh = rng.normal(size=(n + me + 3, bp))
        layout = dict(n=n, me=me, y_eq=n, hess=0, jac=nh, obj_row=n + me)
        outs = []
        for mode in ('numpy', 'torch', 'hip'):
            model = bg.BurgersDeviceModel(nlps, bp)
            if mode == 'numpy':
                W, src, data = Wh.copy(), np.full((nh + nj + 2, bp), 7.0), np.full((n + me + 1, bp), 7.0)
                model._evaluate(W, src, data, layout)
                outs.append((src, data))
                continue
            W = torch.from_numpy(Wh).cuda()
            src = torch.full((nh + nj + 2, bp), 7.0, dtype=torch.float64, device='cuda')
            data = torch.full((n + me + 1, bp), 7.0, dtype=torch.float64, device='cuda')
            (model._evaluate if mode == 'torch' else model.evaluate)(W, src, data, layout)
            torch.cuda.synchronize()
            outs.append((src.cpu().numpy(), data.cpu().numpy()))
        (sn, dn), (st, dt_), (sh, dh) = outs
        assert np.array_equal(sh, sn) and np.array_equal(dh[:n + me], dn[:n + me])
        assert np.allclose(dh[n + me], dn[n + me], rtol=1e-13, atol=0.0)
        assert np.allclose(sh, st, rtol=1e-12, atol=1e-12) and np.allclose(dh, dt_, rtol=1e-12, atol=1e-12)
