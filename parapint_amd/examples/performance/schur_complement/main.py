"""Counterpart of the reference's performance harness (parapint/examples/performance/schur_complement/main.py:
helper :33-60, run :63-139) for the HIP solver: same generator stream, same phases, same table.

    python -m parapint_amd.examples.performance.schur_complement.main --method ssc --n_blocks 4
    python -m torch.distributed.run --nproc-per-node 4 -m parapint_amd.examples.performance.schur_complement.main \\
        --method psc --n_blocks 4

Methods: ``fs`` full-space factorisation of the assembled KKT matrix (HipLDLInterface, one matrix), ``ssc`` serial
Schur complement, ``psc`` parallel Schur complement (one process per GPU, torch.distributed / RCCL).  The sub-solver
choice of the reference (--linear_solver ma27/scipy) has no counterpart: the block factorisations are the batched
LDL^T kernels of libparapint_hip.so.  Defaults are the reference's (n_q_per_block 5000, n_y_multiplier 120,
n_theta 10, 3 nonzeros per row of A); the known answer of its test (examples/tests/test_examples.py:76-99) is
``--n_blocks 3 --n_q_per_block 500 --n_y_multiplier 12`` -> Est Err 0.3163456780."""
import argparse
import os
import time

import numpy as np

from parapint_amd.examples.performance.schur_complement.synthetic_kkt import SyntheticKKT, distribute_blocks
from parapint_amd.linalg.comm import SerialComm


class Result(object):
    def __init__(self):
        self.max_err = None
        self.symbolic_time = None
        self.numeric_time = None
        self.back_solve_time = None
        self.total_time = None


def helper(m, solver, comm, full_space=False):
    """Times the three phases with a barrier in front and the maximum over ranks behind (main.py:33-60)."""
    if full_space:
        kkt = m.build_kkt().tocoo()
        rhs = m.build_rhs()
    else:
        kkt = m.build_kkt(comm=None if comm.size == 1 else comm)
        rhs = m.build_rhs(comm=None if comm.size == 1 else comm)
    comm.barrier()
    t0 = time.time()
    solver.do_symbolic_factorization(kkt)
    t1 = time.time()
    solver.do_numeric_factorization(kkt)
    t2 = time.time()
    x = solver.do_back_solve(rhs)
    t3 = time.time()
    res = Result()
    res.max_err = m.check_result(x, comm=comm)
    times = np.array([t1 - t0, t2 - t1, t3 - t2, t3 - t0])
    if comm.size > 1:
        times = comm.allreduce_max(times)
    res.symbolic_time, res.numeric_time, res.back_solve_time, res.total_time = (float(t) for t in times)
    return res


def run(args):
    from parapint_amd.linalg.hip_schur_complement import HipLDLInterface, HipSchurComplementLinearSolver
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.method != 'psc' and world != 1:
        raise RuntimeError('running serial code with multiple processes')
    if world > 1:
        import torch
        import torch.distributed as dist
        from parapint_amd.linalg.comm import TorchComm
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=torch.device('cuda', torch.cuda.current_device()))
        comm = TorchComm()
    else:
        comm = SerialComm()
    n_blocks = args.n_blocks
    if world > n_blocks:
        raise ValueError('more processes than blocks (mpi_sc_ip_interface.py:322-323)')
    local = distribute_blocks(n_blocks, comm.rank, comm.size)
    m = SyntheticKKT(n_blocks, args.n_q_per_block, args.n_y_multiplier, args.n_theta, args.A_nnz_per_row,
                     local_blocks=local if args.method == 'psc' else None)
    # main.py:75-83: the sub-solver is picked by class.  All three names construct the batched HIP factorisation; the
    # class decides the single-matrix semantics (MA27: cntl(1) = 1e-6, inertia (n - neg, neg, 0); MUMPS: null pivots in
    # the inertia; SciPy: no inertia unless asked)
    from parapint_amd.linalg import InteriorPointMA27Interface, MumpsInterface, ScipyInterface
    if args.subproblem_solver == 'ma27':
        linear_solver_class, linear_solver_options = InteriorPointMA27Interface, dict(cntl_options={1: 1e-6})
    elif args.subproblem_solver == 'mumps':
        linear_solver_class, linear_solver_options = MumpsInterface, dict()
    else:
        linear_solver_class, linear_solver_options = ScipyInterface, dict(compute_inertia=False)
    if args.method == 'fs':
        solver = linear_solver_class(**linear_solver_options)
    else:
        # (the block factorisations are one batch on the GPU: the per-block sub-solver objects of the reference's
        # constructor are accepted and not called; the pivot tolerance of the chosen class applies to the batch)
        tol = linear_solver_options.get('cntl_options', {}).get(1)
        solver = HipSchurComplementLinearSolver({i: None for i in local}, None, comm=comm,
                                                symbolic_pivot_threshold=None if tol is None else max(tol, 0.01))
    res = helper(m, solver, comm, full_space=(args.method == 'fs'))
    method_map = {'fs': 'Full Space', 'ssc': 'Serial Schur-Complement', 'psc': 'Parallel Schur-Complement'}
    if comm.rank == 0:
        head = ['method', '# processes', '# blocks', 'n_q_per_block', 'n_y_multiplier', 'n_theta', 'A NNZ per row',
                'Est Err', 'Symb Fact (s)', 'Num Fact (s)', 'Back Solve (s)', 'Total Time (s)']
        vals = [method_map[args.method], comm.size, n_blocks, args.n_q_per_block, args.n_y_multiplier, args.n_theta,
                args.A_nnz_per_row, '%.10f' % res.max_err, '%.4f' % res.symbolic_time, '%.4f' % res.numeric_time,
                '%.4f' % res.back_solve_time, '%.4f' % res.total_time]
        print(''.join(('%-30s' if i == 0 else '%-15s') % h for i, h in enumerate(head)))
        print(''.join(('%-30s' if i == 0 else '%-15s') % str(v) for i, v in enumerate(vals)))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return res


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--method', type=str, required=True, choices=['fs', 'ssc', 'psc'],
                        help='fs: full space, ssc: serial Schur complement, psc: parallel Schur complement')
    parser.add_argument('--n_blocks', type=int, required=True)
    parser.add_argument('--subproblem_solver', '--linear_solver', type=str, default='ma27', choices=['ma27', 'mumps', 'scipy'],
                        help='sub-solver class, as main.py:75-83 (all are served by the HIP factorisation)')
    parser.add_argument('--n_q_per_block', type=int, default=5000)
    parser.add_argument('--n_y_multiplier', type=int, default=120)
    parser.add_argument('--n_theta', type=int, default=10)
    parser.add_argument('--A_nnz_per_row', type=int, default=3)
    return parser.parse_args(argv)


if __name__ == '__main__':
    run(parse_args())
