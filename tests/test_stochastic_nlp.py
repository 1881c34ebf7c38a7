"""Nonlinear scenario problems through the stochastic Schur-complement interface: the host interface over NLP objects
against the device-resident producer with a device model (DeviceStochasticNLPInterface), on numpy engines and on the device."""
import numpy as np
import pytest

from parapint_amd.examples import stochastic_nlp as ex


def _host_loop(nlps, fs, solver):
    from parapint_amd.algorithms.interior_point import IPOptions, InteriorPointStatus, ip_solve
    from parapint_amd.interfaces.schur_complement.sc_ip_interface import StochasticSchurComplementInteriorPointInterface
    it = StochasticSchurComplementInteriorPointInterface(nlps, fs)
    opt = IPOptions()
    opt.linalg.solver = solver
    assert ip_solve(it, opt) == InteriorPointStatus.optimal
    return it


def _device_loop(nlps, fs, engine):
    from parapint_amd.algorithms.device_interior_point import ip_solve_device
    from parapint_amd.algorithms.interior_point import IPOptions, InteriorPointStatus
    from parapint_amd.interfaces.schur_complement.device_sc_ip_interface import DeviceStochasticNLPInterface
    from parapint_amd.linalg.comm import SerialComm
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
    it = DeviceStochasticNLPInterface(nlps, fs, ex.ScenarioDeviceModel)
    opt = IPOptions()
    opt.linalg.solver = HipSchurComplementLinearSolver({i: None for i in range(len(nlps))}, None, comm=SerialComm(), engine=engine,
                                                       result_buffers=0 if engine is not None else 2)
    hist = []
    status, _ = ip_solve_device(it, opt, history=hist)
    assert status == InteriorPointStatus.optimal
    return it, hist


def _oracle_solver(n):
    from oracle.schur_complement import SchurComplementLinearSolver as OracleSC
    from oracle.subsolvers import ScipyInterface as OracleScipy
    return OracleSC({i: OracleScipy(compute_inertia=True) for i in range(n)}, OracleScipy(compute_inertia=True))


def _same(it, host, n, tol):
    assert abs(it.evaluate_objective() - host.evaluate_objective()) <= tol
    assert np.abs(it.first_stage_solution() - np.asarray(host.get_primals().get_block(n))).max() <= tol
    for i in range(n):
        assert np.abs(it.scenario_primals(i) - np.asarray(host.get_primals().get_block(i))).max() <= tol


def test_nonlinear_scenarios_on_cpu_engines_match_the_host_loop():
    from hostsim_engine import HostSimDeviceEngine
    nlps, fs = ex.random_scenarios(6, seed=3)
    it, hist = _device_loop(nlps, fs, HostSimDeviceEngine())
    nlps2, _ = ex.random_scenarios(6, seed=3)
    host = _host_loop(nlps2, fs, _oracle_solver(6))
    _same(it, host, 6, 1e-6)
    y = np.concatenate([it.scenario_primals(i)[3:] for i in range(6)])
    assert (np.abs(y) <= 1.5 + 1e-7).all() and np.abs(y).max() > 1.49              # a bound on the recourse variables is active


@pytest.mark.gpu
def test_nonlinear_scenarios_on_the_device():
    from hostsim_engine import HostSimDeviceEngine
    nlps, fs = ex.random_scenarios(200, n_f=5, n_y=23, seed=8)
    it, hist = _device_loop(nlps, fs, None)
    nlps2, _ = ex.random_scenarios(200, n_f=5, n_y=23, seed=8)
    ref, ref_hist = _device_loop(nlps2, fs, HostSimDeviceEngine())
    assert len(hist) == len(ref_hist)
    for a, b in zip(hist, ref_hist):
        assert np.allclose(a[:4], b[:4], rtol=1e-6, atol=2e-9), (a, b)
    assert np.abs(it.first_stage_solution() - ref.first_stage_solution()).max() <= 1e-7
