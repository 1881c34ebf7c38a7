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
