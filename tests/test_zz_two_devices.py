"""Runs LAST (file order) and only where two HIP devices are visible: `python bench.py --gpus 2` with one GPU per rank
over RCCL -- the first configuration of the multi-GPU product.  On the one-GPU boxes this suite normally runs on it is
skipped; tests/test_hip_solver.py::test_bench_two_ranks_started_by_the_script_itself rehearses the same code path with two
ranks on one device over gloo, tests/test_hip_solver.py::test_rccl_collectives_on_solver_buffers the library's RCCL calls in
a one-rank group."""
import pytest

pytestmark = pytest.mark.gpu


def _two_devices():
    import torch
    return torch.cuda.device_count() >= 2


def test_bench_two_ranks_one_gpu_each_over_rccl():
    if not _two_devices():
        pytest.skip('one HIP device')
    from test_hip_solver import _run_bench
    res = _run_bench({}, '--gpus', '2', '--workload', 'C2', '--steps', '4', '--warmup', '2', '--no-cpu-baseline',
                     '--no-boundary', '--no-ip-loop', '--profile-steps', '1')
    assert res['n_gpus'] == 2 and res['correct'] is True and res['scaling'] == 'strong'
    assert res['config']['world_size'] == 2 and res['config']['blocks_per_gpu'] == 32
    assert res['residual'] <= 1e-8 and res['inertia'] == res['expected_inertia']
    assert res['solution_check']['on'] is True and res['solution_check']['backward_error_last_step'] <= 1e-10
    assert res['rccl_ranks'] == 2      # (the library's own communicator: default for two or more ranks)
    # every rank holding the workload's full block count
    res = _run_bench({}, '--gpus', '2', '--workload', 'C2', '--steps', '4', '--warmup', '2', '--no-cpu-baseline',
                     '--no-boundary', '--no-ip-loop', '--profile-steps', '1', '--scaling', 'weak')
    assert res['correct'] is True and res['scaling'] == 'weak' and res['config']['blocks_per_gpu'] == 64
    assert res['residual'] <= 1e-8
