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
