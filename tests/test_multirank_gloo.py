"""world_size-2 run of the multi-rank host path on CPU (gloo), one process per rank, launched
the way the driver launches bench.py."""
import os
import socket
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_gloo_schur_solver():
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(HERE, 'multirank_worker.py')]
    env = dict(os.environ)
    env['OMP_NUM_THREADS'] = '1'
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    text = out.stdout.decode()
    assert out.returncode == 0, text[-4000:]
    assert 'rank 0 ok' in text and 'rank 1 ok' in text


import pytest  # noqa: E402


@pytest.mark.parametrize('world', [3, 4])
def test_reference_mpi_8x8_on_three_and_four_ranks(world):
    """The reference's MPI test of the path runs on 1-4 ranks (test_mpi_explicit_schur_complement.py, `all_proc`): three
    blocks on three ranks, and on four -- where the last rank owns no block at all and still takes part in every collective."""
    _run_mpi8(world)


@pytest.mark.gpu
def test_reference_mpi_8x8_on_four_ranks_sharing_the_device():
    """The same through the product engine: four processes on the one device (gloo between them), the last rank without a
    block -- its handle holds no pattern group, its S contribution is zero, it still solves for the coupling variables."""
    _run_mpi8(4, '--gpu')


def _run_mpi8(world, *args):
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.join(HERE, 'mpi8_worker.py')] + list(args)
    env = dict(os.environ)
    env['OMP_NUM_THREADS'] = '1'
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    text = out.stdout.decode()
    assert out.returncode == 0, text[-4000:]
    assert all('rank %d of %d ok' % (r, world) in text for r in range(world))
