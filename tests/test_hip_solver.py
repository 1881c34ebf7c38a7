"""GPU parity suite: the product path (HIP kernels through the C ABI) against the reference's
golden vectors, the oracle and dense algebra.  Run on the GPU box with ``pytest -m gpu``."""
import numpy as np
import pytest

import solver_cases as sc

pytestmark = pytest.mark.gpu


def make_engine():
    return None          # product default: HipEngine (raises without a GPU / the HIP library)


def test_native_library_is_the_one_in_tree():
    from parapint_amd import _native
    lib = _native.load_library()
    assert _native.LIB_PATH.endswith('parapint_amd/csrc/libparapint_hip.so')
    assert lib is not None


def test_sub_solver_contract(golden):
    sc.case_sub_solver_contract(make_engine, golden)


@pytest.mark.parametrize('mpi', [False, True])
def test_bordered_8x8(golden, mpi):
    sc.case_bordered_8x8(make_engine, golden, mpi)


@pytest.mark.parametrize('shape', [(3, 20, 2, 4), (4, 50, 3, 6)])
def test_small_synthetic(golden, shape):
    sc.case_small_synthetic(make_engine, golden, shape)


def test_known_answer(golden):
    sc.case_known_answer(make_engine, golden)


def test_oracle_schur_and_solution():
    sc.case_oracle_schur(make_engine, (5, 40, 2, 8))
    sc.case_oracle_schur(make_engine, (70, 30, 2, 10))     # more than one 64-instance wave, ragged tail


def test_heterogeneous_groups():
    sc.case_heterogeneous(make_engine)


def test_error_behaviour():
    sc.case_errors(make_engine)


def test_config2_against_full_space_superlu():
    # BASELINE.json configs[1]: 64 scenarios x 2k primal vars/block, 100 coupling vars
    solver, model = sc.case_against_oracle(make_engine, (64, 400, 4, 100), iteration=1)
    st = solver.plan_stats[0]
    assert st['n'] == 3700 and st['batch'] == 64


def test_refactor_with_new_values_same_object():
    from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT
    from parapint_amd.linalg.comm import SerialComm
    model = SyntheticKKT(10, 50, 3, 10)
    solver = sc.new_solver(make_engine, 10)
    rhs = model.build_rhs(comm=SerialComm())
    solver.do_symbolic_factorization(model.build_kkt(comm=SerialComm()))
    for it in (0, 1, 2):
        kkt = model.build_kkt(comm=SerialComm(), iteration=it)
        solver.do_numeric_factorization(kkt)
        x = solver.do_back_solve(rhs)
        assert sc.scaled_residual(kkt.tocoo(), x.flatten(), rhs.flatten()) <= sc.RESID_TOL


def test_full_size_blocks_property():
    # configuration-3 block shape (n_i = 9200, n_c = 200) on 130 instances: residual + inertia properties
    solver, model = sc.case_against_oracle(make_engine, (130, 1000, 4, 200), iteration=4, check_full_space=False)
    assert solver.plan_stats[0]['n'] == 9200
