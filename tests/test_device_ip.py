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
    host_kkt = it.host.evaluate_primal_dual_kkt_matrix()      # (dk itself carries scenario 0's blocks for every scenario)
    src, coef = it._value_map
    N = len(qps)
    for ndx in range(N):
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
    assert dk.nsrc == it.nsrc and len(src) == ref.size


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
    assert len(rows) == len(hist)                                # same number of iterations
    for r, h in zip(rows, hist):
        # the host loop prints 3 significant digits: primal / dual / complementarity infeasibility and the barrier
        # (below 1e-9 the measures are rounding noise of two different summation orders)
        for a, b in zip(r[2:6], (h[0], h[1], h[2], h[3])):
            assert abs(a - b) <= 6e-3 * max(abs(a), abs(b)) + 2e-9
    zh = np.asarray(hi.get_primals().get_block(n_scenarios))
    assert np.abs(it.first_stage_solution() - zh).max() <= 1e-7 * max(1.0, np.abs(zh).max())
    for ndx in (0, n_scenarios - 1):
        xh = hi.scenario_interface(ndx).get_primals()
        assert np.abs(it.scenario_primals(ndx) - xh).max() <= 1e-6 * max(1.0, np.abs(xh).max())


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
