"""The known answers the reference's own tests hold for the time-block (dynamic) Schur-complement interface
(parapint/interfaces/schur_complement/tests/test_sc_ip_interface.py:150-450 and, with MPI containers,
test_mpi_sc_ip_interface.py:164-486), re-expressed for the restated interface over QuadraticProgram time blocks:
6 finite elements in 3 time blocks, the control constant over 2, p <= 1.75 as a bound, a hand-set state; then the solves
of :454-572 (one Newton step solves the equality-constrained problem; the interior-point loop with bounds / inequalities
gives the full-space optimum)."""
import math

import numpy as np
import pytest

from parapint_amd.examples import dynamics as dy
from parapint_amd.sparse.block_containers import BlockVector

TS, BARRIER = 1.0, 0.1


def _bv(blocks):
    v = BlockVector(len(blocks))
    for i, b in enumerate(blocks):
        v.set_block(i, b if hasattr(b, 'get_block') else np.asarray(b, dtype=np.double))
    return v


def make_interface(comm=None):
    """setUpClass of the reference's TestSCIPInterface (:152-247): x[t] = t, p = 0.5 / 1 / 1.5, duals of the block's own
    constraints 1..6, link duals = block index, upper-bound dual of p = block index, coupling states 3 and 6.  With a
    communicator: the MPI flavour (test_mpi_sc_ip_interface.py:164-293), every rank sets the blocks it owns."""
    it = dy.Problem(num_finite_elements=6, constant_control_duration=2, time_scale=TS, num_time_blocks=3, p_ub=1.75, comm=comm)
    if comm is not None and comm.size > 1:
        return _set_state_distributed(it)
    T = 3
    primals = [[0, 1, 2, 0.5], [2, 3, 4, 1], [4, 5, 6, 1.5], [3, 6]]
    own = [[1, 2], [3, 4], [5, 6]]
    duals_eq = []
    for ndx in range(T):
        back = np.zeros(0) if ndx == 0 else np.ones(1) * ndx
        fwd = np.zeros(0) if ndx == T - 1 else np.ones(1) * ndx
        duals_eq.append(_bv([own[ndx], back, fwd]))
    it.set_primals(_bv(primals))
    it.set_duals_eq(_bv(duals_eq))
    it.set_duals_ineq(_bv([np.zeros(0)] * T))
    it.set_duals_slacks_lb(_bv([np.zeros(0)] * T))
    it.set_duals_slacks_ub(_bv([np.zeros(0)] * T))
    it.set_duals_primals_lb(_bv([np.zeros(4)] * T + [np.zeros(2)]))
    it.set_duals_primals_ub(_bv([[0, 0, 0, ndx] for ndx in range(T)] + [np.zeros(2)]))
    it.set_barrier_parameter(BARRIER)
    return it


@pytest.fixture(scope='module')
def interface():
    return make_interface()


def _set_state_distributed(it):
    """The same state through the rank-distributed containers: a rank sets the blocks it owns and the (replicated)
    coupling block."""
    T = 3
    primals = [[0, 1, 2, 0.5], [2, 3, 4, 1], [4, 5, 6, 1.5], [3, 6]]
    own = [[1, 2], [3, 4], [5, 6]]

    def vec(extra, per_block, last=None):
        v = it._vector(extra)
        for ndx in it.local_block_indices:
            b = per_block(ndx)
            v.set_block(ndx, b if hasattr(b, 'get_block') else np.asarray(b, dtype=np.double))
        if extra:
            v.set_block(T, np.asarray(last, dtype=np.double))
        return v
    it.set_primals(vec(True, lambda n: primals[n], primals[T]))
    it.set_duals_eq(vec(False, lambda n: _bv([own[n], np.zeros(0) if n == 0 else np.ones(1) * n,
                                              np.zeros(0) if n == T - 1 else np.ones(1) * n])))
    for name in ('duals_ineq', 'duals_slacks_lb', 'duals_slacks_ub'):
        getattr(it, 'set_' + name)(vec(False, lambda n: np.zeros(0)))
    it.set_duals_primals_lb(vec(True, lambda n: np.zeros(4), np.zeros(2)))
    it.set_duals_primals_ub(vec(True, lambda n: [0, 0, 0, n], np.zeros(2)))
    it.set_barrier_parameter(BARRIER)
    return it


def expected_rhs():
    """:402-450, in the flat order [block 0 | block 1 | block 2 | coupling block]."""
    s, b = (lambda t: math.sin(TS * t) + 1), BARRIER
    return -np.array([
        1 * (0 - s(0)) + (-1) * 1, 2 * (1 - s(1)) + 2 * 1 + (-1) * 2, 1 * (2 - s(2)) + 2 * 2 + 1 * 0,
        0 + (-1) * 1 + (-1) * 2 + b / (1.75 - 0.5), 1 - (0 + (0.5 - 1)), 2 - (1 + (0.5 - 2)),
        1 * (2 - s(2)) + (-1) * 3 + 1 * 1, 2 * (3 - s(3)) + 2 * 3 + (-1) * 4, 1 * (4 - s(4)) + 2 * 4 + 1 * 1,
        0 + (-1) * 3 + (-1) * 4 + b / (1.75 - 1.0), 3 - (2 + (1 - 3)), 4 - (3 + (1 - 4)), 2 - 3,
        1 * (4 - s(4)) + (-1) * 5 + 1 * 2, 2 * (5 - s(5)) + 2 * 5 + (-1) * 6, 1 * (6 - s(6)) + 2 * 6,
        0 + (-1) * 5 + (-1) * 6 + b / (1.75 - 1.5), 5 - (4 + (1.5 - 5)), 6 - (5 + (1.5 - 6)), 4 - 6,
        2 - 3, 4 - 6, 0 + (-1) * 1 + (-1) * 0, 0 + (-1) * 2 + (-1) * 1])


def _flat(v):
    return np.asarray(v.flatten(), dtype=np.double)


def test_sizes_bounds_and_state(interface):
    it = interface
    assert it.n_primals() == 14 and it.n_eq_constraints() == 10 and it.n_ineq_constraints() == 0          # :249, 306, 309
    assert np.all(np.isneginf(_flat(it.primals_lb())))                                                       # :252-256
    inf = np.inf
    assert np.allclose(_flat(it.primals_ub()), [inf, inf, inf, 1.75, inf, inf, inf, 1.75, inf, inf, inf, 1.75, inf, inf])
    assert np.allclose(_flat(it.get_primals()), [0, 1, 2, 0.5, 2, 3, 4, 1, 4, 5, 6, 1.5, 3, 6])             # :263-266
    assert np.allclose(_flat(it.get_duals_eq()), [1, 2, 0, 3, 4, 1, 1, 5, 6, 2])                             # :322-325
    assert np.allclose(_flat(it.get_duals_primals_lb()), np.zeros(14))
    assert np.allclose(_flat(it.get_duals_primals_ub()), [0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 2, 0, 0])         # :381-384
    for v in (it.ineq_lb(), it.ineq_ub(), it.get_duals_ineq(), it.evaluate_ineq_constraints(), it.get_slacks(),
              it.get_duals_slacks_lb(), it.get_duals_slacks_ub()):
        assert v.nblocks == 3 and v.size == 0                                                                # :312-320, 327-335, ...


def test_objective_and_gradient(interface):
    s = lambda t: math.sin(TS * t) + 1
    # trapezoid: every interior time point of a block counts twice, the shared ones once per block (:268-283)
    expected = sum(0.5 * (a - s(a)) ** 2 + 0.5 * (b - s(b)) ** 2 for a, b in zip(range(0, 6), range(1, 7)))
    assert abs(interface.evaluate_objective() - expected) <= 1e-12
    grad = [1 * (0 - s(0)), 2 * (1 - s(1)), 1 * (2 - s(2)), 0, 1 * (2 - s(2)), 2 * (3 - s(3)), 1 * (4 - s(4)), 0,
            1 * (4 - s(4)), 2 * (5 - s(5)), 1 * (6 - s(6)), 0, 0, 0]                                           # :285-304
    assert np.allclose(_flat(interface.evaluate_grad_objective()), grad)


def test_constraints_and_jacobian(interface):
    eq = [1 - (0 + (0.5 - 1)), 2 - (1 + (0.5 - 2)), 2 - 3,
          3 - (2 + (1 - 3)), 4 - (3 + (1 - 4)), 2 - 3, 4 - 6,
          5 - (4 + (1.5 - 5)), 6 - (5 + (1.5 - 6)), 4 - 6]                                                     # :337-349
    assert np.allclose(_flat(interface.evaluate_eq_constraints()), eq)
    #      x0  x1  x2  p0  x2  x3  x4  p2  x4  x5  x6  p4  z0  z1                                             (:357-372)
    jac = [[-1, 2, 0, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
           [0, -1, 2, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
           [0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, -1, 0],
           [0, 0, 0, 0, -1, 2, 0, -1, 0, 0, 0, 0, 0, 0],
           [0, 0, 0, 0, 0, -1, 2, -1, 0, 0, 0, 0, 0, 0],
           [0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, -1, 0],
           [0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, -1],
           [0, 0, 0, 0, 0, 0, 0, 0, -1, 2, 0, -1, 0, 0],
           [0, 0, 0, 0, 0, 0, 0, 0, 0, -1, 2, -1, 0, 0],
           [0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, -1]]
    assert np.allclose(interface.evaluate_jacobian_eq().toarray(), jac)
    assert interface.evaluate_jacobian_ineq().shape == (0, 14)                                                 # :374-378


def test_primal_dual_kkt_rhs(interface):
    """:402-450: per block [grad L of its variables | its constraints | backward link], then the coupling block
    [forward links of blocks 0, 1 | grad L of the coupling states]."""
    expected = expected_rhs()
    got = _flat(interface.evaluate_primal_dual_kkt_rhs())
    assert got.size == 24 and np.allclose(got, expected)


# ---- the solves of TestSCIPInterfaceWithSolve (:454-572); where the reference compares with ipopt on the full-horizon
# model, the comparison here is the full-horizon QP solved directly (no bounds: one linear solve) or by the same loop
def _full_horizon(**kw):
    return dy.build_time_block(t0=0, delta_t=1, num_finite_elements=90, constant_control_duration=10, time_scale=0.1, **kw)


def _trajectories(it):
    x, p = {}, {}
    for ndx in range(3):
        for t, v in it.x(ndx).items():
            assert t not in x or abs(x[t] - v) <= 1e-7          # the shared time points agree (:504-507)
            x[t] = v
        p.update(it.p(ndx))
    return x, p


def _oracle_sc():
    from oracle.schur_complement import SchurComplementLinearSolver as OracleSC
    from oracle.subsolvers import ScipyInterface as OracleScipy
    return OracleSC({i: OracleScipy(compute_inertia=True) for i in range(3)}, OracleScipy(compute_inertia=True))


def test_one_newton_step_solves_the_equality_constrained_problem():
    """:455-520: from the initial point with barrier 0, one solve of the KKT system is the optimum."""
    from oracle.subsolvers import ScipyInterface as OracleScipy
    it = dy.Problem(p_ub=None)
    for name in ('primals', 'slacks', 'duals_eq', 'duals_ineq', 'duals_primals_lb', 'duals_primals_ub',
                 'duals_slacks_lb', 'duals_slacks_ub'):
        getattr(it, 'set_' + name)(getattr(it, 'init_' + name)())
    it.set_barrier_parameter(0)
    kkt, rhs = it.evaluate_primal_dual_kkt_matrix(), it.evaluate_primal_dual_kkt_rhs()
    solver = OracleScipy()
    solver.do_symbolic_factorization(kkt)
    solver.do_numeric_factorization(kkt)
    it.set_primal_dual_kkt_solution(solver.do_back_solve(rhs))
    new = it.get_primals().copy_structure()
    for b in range(4):
        new.set_block(b, np.asarray(it.get_primals().get_block(b)) + np.asarray(it.get_delta_primals().get_block(b)))
    it.set_primals(new)
    x, p = _trajectories(it)
    qp, x_times, p_times = _full_horizon(p_ub=None)
    n, me = qp.n, qp.A_eq.shape[0]
    H = (qp.H + qp.H.T).toarray() - np.diag(qp.H.diagonal())
    K = np.block([[H, qp.A_eq.toarray().T], [qp.A_eq.toarray(), np.zeros((me, me))]])
    sol = np.linalg.solve(K, np.concatenate([-qp.c, qp.b_eq]))[:n]
    for i, t in enumerate(x_times):
        assert abs(x[t] - sol[i]) <= 1e-7
    for i, t in enumerate(p_times):
        assert abs(p[t] - sol[len(x_times) + i]) <= 1e-7


@pytest.mark.parametrize('kw', [dict(p_ub=None), dict(p_ub=1.75), dict(p_ub=None, p_ub_as_constraint=1.75)])
def test_interior_point_loop_gives_the_full_horizon_optimum(kw):
    """:522-572 (_ip_helper, without / with bounds / with inequalities, over the Schur-complement solver)."""
    from parapint_amd.algorithms.interior_point import IPOptions, InteriorPointStatus, ip_solve
    from parapint_amd.interfaces.schur_complement.sc_ip_interface import StochasticSchurComplementInteriorPointInterface
    from oracle.schur_complement import SchurComplementLinearSolver as OracleSC
    from oracle.subsolvers import ScipyInterface as OracleScipy
    it = dy.main(_oracle_sc(), **kw)
    x, p = _trajectories(it)
    qp, x_times, p_times = _full_horizon(**kw)
    mono = StochasticSchurComplementInteriorPointInterface([qp], [[0]])
    opt = IPOptions()
    opt.linalg.solver = OracleSC({0: OracleScipy(compute_inertia=True)}, OracleScipy(compute_inertia=True))
    assert ip_solve(mono, opt) == InteriorPointStatus.optimal
    sol = np.asarray(mono.get_primals().get_block(0))
    assert max(abs(x[t] - sol[i]) for i, t in enumerate(x_times)) <= 1e-6
    assert max(abs(p[t] - sol[len(x_times) + i]) for i, t in enumerate(p_times)) <= 1e-6
    if kw.get('p_ub') or kw.get('p_ub_as_constraint'):
        assert abs(max(p.values()) - 1.75) <= 1e-6                  # the limit on the control is active
