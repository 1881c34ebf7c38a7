"""CPU suite: the solver class's host logic driven through the TEST-ONLY host-interpreter engine
(the HIP kernels execute the same plan on the GPU; see tests/test_hip_solver.py for parity on device)."""
import pytest

import solver_cases as sc
from hostsim_engine import HostSimBoundaryEngine, HostSimEngine


def make_engine():
    return HostSimEngine()


def test_sub_solver_contract(golden):
    sc.case_sub_solver_contract(make_engine, golden)


@pytest.mark.parametrize('mpi', [False, True])
def test_bordered_8x8(golden, mpi):
    sc.case_bordered_8x8(make_engine, golden, mpi)


@pytest.mark.parametrize('mpi', [False, True])
def test_bordered_8x8_with_the_unsymmetric_blocks_of_the_reference_tests(golden, mpi):
    sc.case_bordered_8x8_original(make_engine, golden, mpi)


def test_unsymmetric_blocks_through_the_scipy_interface_route():
    sc.case_general_blocks_random(make_engine)


@pytest.mark.parametrize('shape', [(3, 20, 2, 4), (4, 50, 3, 6)])
def test_small_synthetic(golden, shape):
    sc.case_small_synthetic(make_engine, golden, shape)


def test_known_answer(golden):
    sc.case_known_answer(make_engine, golden)


@pytest.mark.parametrize('shape', [(1, 10, 2, 1), (1, 5, 2, 5), (65, 6, 2, 2)])
def test_edge_shapes(shape):
    sc.case_against_oracle(make_engine, shape, iteration=1)


def test_against_oracle_small():
    sc.case_against_oracle(make_engine, (6, 60, 3, 12), iteration=2)
    sc.case_oracle_schur(make_engine, (5, 40, 2, 8))


def test_heterogeneous_groups():
    sc.case_heterogeneous(make_engine)


def test_error_behaviour():
    sc.case_errors(make_engine)


def test_singular_schur_complement_is_a_status_when_asked_for_one():
    sc.case_singular_schur_complement(make_engine)


def test_nested_block_matrices_and_vectors():
    sc.case_nested_blocks(make_engine)


def test_inertia_correction_pattern_growth():
    sc.case_inertia_correction_pattern_growth(make_engine)


def test_pivot_order_refresh_after_static_breakdown():
    sc.case_pivot_order_refresh(make_engine)


def test_instances_of_one_group_that_need_different_pivot_sequences():
    """Round 5: the group is split into variants instead of reporting a regular matrix singular (ma27_interface.py:110-140)."""
    sc.case_conflicting_pivots(make_engine)


def test_ip_solve_call_pattern():
    sc.case_ip_solve_call_pattern(make_engine)


def test_memory_reallocation_retry_loop():
    sc.case_reallocation(make_engine, lambda solver: solver._eng.required_bytes())


def test_status_severity_order():
    sc.case_status_severity()


def test_pivot_growth_guard():
    sc.case_growth_guard(make_engine)


@pytest.mark.parametrize('shape', [(2, 3), (6, 4), (9, 5)])
def test_dynamic_time_blocks_with_local_coupling_maps(shape):
    sc.case_dynamic(make_engine, shape[0], shape[1])


def test_dynamic_block_tridiagonal_plumbing_on_the_host():
    """The block-tridiagonal path of the host class (RCM ordering, padding, flat D | E layouts of S and Q, inertia
    without the padding rows) with the interpreter keeping S dense in the permuted ordering."""
    solver, model = sc.case_dynamic(make_engine, 12, 4, expect_block_tridiagonal=True, dense_limit=8)
    gs, G = solver._btd
    assert G >= 3 and gs * G >= model.n_coupling


@pytest.mark.parametrize('dense_limit', [None, 8])
def test_dynamic_problem_through_the_inertia_correction_loop(dense_limit):
    solver = sc.case_dynamic_regularised(make_engine, dense_limit)
    assert (solver._btd is not None) == (dense_limit is not None)


def test_reference_realloc_matrix_through_the_sub_solver_adapters():
    """linalg/tests/test_realloc.py:10-61 (the reference runs it on MUMPS only) on the host interpreter, 2000 rows."""
    sc.case_reference_realloc_matrix(make_engine, n=2000)


def test_harness_sub_solver_switch():
    """examples/performance/schur_complement/main.py:75-83: the sub-solver is picked by name."""
    from parapint_amd.examples.performance.schur_complement.main import parse_args
    assert parse_args(['--method', 'fs', '--n_blocks', '3']).subproblem_solver == 'ma27'
    assert parse_args(['--method', 'psc', '--n_blocks', '3', '--subproblem_solver', 'mumps']).subproblem_solver == 'mumps'
    assert parse_args(['--method', 'ssc', '--n_blocks', '3', '--linear_solver', 'scipy']).subproblem_solver == 'scipy'


def test_host_boundary_fast_paths():
    sc.case_boundary_fast_paths(HostSimBoundaryEngine)


def test_entries_declared_constant_by_the_producer():
    sc.case_constant_entries(HostSimBoundaryEngine)


def test_flat_value_vectors_over_the_symbolic_pattern():
    sc.case_flat_values(HostSimBoundaryEngine)


def test_host_engine_refuses_a_subset_of_runs_for_a_row_that_is_not_whole():
    """The interpreter keeps its own record of which staging rows hold every entry of their block (what the library's
    staging rows are to the product): a solver whose flags claim more than that -- here set by hand -- is caught when it
    hands over the runs of the variable entries only."""
    import numpy as np
    from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
    from parapint_amd.linalg.comm import SerialComm
    N = 4
    model = SyntheticKKT(N, 3, 8, 2)
    comm = SerialComm()
    solver = sc.new_solver(HostSimBoundaryEngine, N)
    solver.do_symbolic_factorization(model.build_kkt(comm=comm, iteration=0))
    solver.declare_constant_entries(model.constant_entries())
    k = model.build_kkt(comm=comm, iteration=0)
    solver.do_numeric_factorization(k)                      # (first pass: index arrays compared, rows staged whole)
    solver.do_numeric_factorization(k)
    whole = [rec[1].copy() for rec in solver._eng._rows_whole.values()]
    assert whole and all(w.all() for w in whole)
    for g in solver._groups:                                # a new staging array whose rows the solver wrongly takes for whole
        g._staging = None
        _ = g.staging
        g.full_rows[:] = True
    try:
        solver.do_numeric_factorization(k)
        raise AssertionError('the subset of runs went into rows that hold nothing')
    except AssertionError as e:
        assert 'does not hold every entry' in str(e)


def test_bench_host_boundary_section_on_the_host_engine():
    """bench.py's host-boundary legs (COO blocks; constant entries declared; flat value vectors, with and without the
    declaration) with the solver class on the host-simulation engine: every leg returns a rate, a residual and the phases."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
    from parapint_amd.linalg.comm import SerialComm
    N = 6
    model = SyntheticKKT(N, 3, 8, 2)
    comm = SerialComm()
    solver = sc.new_solver(HostSimBoundaryEngine, N)

    def residual_check(kkt, x, rhs):
        return sc.scaled_residual(kkt.toarray(), x.flatten(), rhs.flatten())
    expected = (N * (model.n_y + model.n_q) + model.n_theta, N * (model.n_y + model.n_theta), 0)
    boundary, declared, flat_plain, resid, ok, t_sym = bench.host_boundary_section(solver, model, comm, 1, None, 3,
                                                                                  residual_check, expected)
    assert ok and resid <= 1e-10 and t_sym > 0
    for leg in (boundary, declared, flat_plain, flat_plain['constant_declared']):
        assert leg['it_per_s'] > 0 and 'values to device' in leg['phases_ms'] and 'solve' in leg['phases_ms']
    assert declared['residual'] <= 1e-10 and flat_plain['residual'] <= 1e-10
    assert all(g.var_runs is None for g in solver._groups)              # (the declaration was withdrawn again)


def test_random_systems_in_random_input_forms():
    """A slice of tools/fuzz_solver.py: random block-bordered systems (several pattern groups, mapped or uniform coupling),
    six factorisations each with the values handed over in changing forms, every solve against dense algebra."""
    import os
    import sys
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools')
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import fuzz_solver
    bad = [r for r in (fuzz_solver.one(seed) for seed in range(120)) if r is not None]
    assert not bad, bad


def test_inaccurate_pivot_sequence_is_refined_host_containers():
    sc.case_refinement_fixture(HostSimEngine)


def test_inaccurate_pivot_sequence_is_refined_device_vectors_blocking_and_deferred():
    from hostsim_engine import HostSimDeviceEngine
    sc.case_refinement_fixture(HostSimDeviceEngine, device_vectors=True)


def test_adversarial_systems_are_never_returned_inaccurate():
    from hostsim_engine import HostSimBoundaryEngine
    sc.case_adversarial_systems(HostSimBoundaryEngine)


def test_zero_pivot_test_inside_a_mixed_scale_block_pivot():
    sc.case_mixed_scale_block_pivot(HostSimEngine)
