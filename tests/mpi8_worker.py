"""Worker of tests/test_multirank_gloo.py::test_reference_mpi_8x8_on_three_and_four_ranks: the reference's own MPI test of the
path (linalg/schur_complement/tests/test_mpi_explicit_schur_complement.py:22-115 -- the 8x8 system with Q = [[0, 0], [0, 1]],
ownership (ndx - rank) % size == 0, run there on 1-4 ranks) on WORLD_SIZE gloo ranks with the TEST-ONLY host interpreter:
with four ranks the last one owns no block at all.  Solution, S and inertia against the golden vectors of the reference's
solver classes; a second numeric factorisation + solve on the same object (:113-115); symmetric variant through the plain
route, the unsymmetric original through ScipyInterface objects (general_blocks.py)."""
import os
import sys

import numpy as np
import torch.distributed as dist
from scipy.sparse import coo_matrix

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from hostsim_engine import HostSimEngine  # noqa: E402
from parapint_amd.linalg import ScipyInterface  # noqa: E402
from parapint_amd.linalg.comm import TorchComm  # noqa: E402
from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver  # noqa: E402
from parapint_amd.linalg.results import LinearSolverStatus  # noqa: E402
from parapint_amd.sparse.block_containers import MPIBlockMatrix, MPIBlockVector  # noqa: E402


GPU = '--gpu' in sys.argv       # the product engine (the ranks share the devices that are there; collectives over gloo)


def main():
    dist.init_process_group('gloo')
    if GPU:
        import torch
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count())
    comm = TorchComm()
    rank, size = comm.rank, comm.size
    golden = np.load(os.path.join(HERE, 'golden', 'reference_vectors.npz'))
    owners = [ndx % size for ndx in range(3)]                  # (ndx - rank) % size == 0
    own = np.array([[-1] * 4 for _ in range(4)])
    for i in range(3):
        own[i, i] = own[3, i] = owners[i]
    mine = [i for i in range(3) if owners[i] == rank]
    border = [np.array([[0, -1], [0, 0.]]), np.array([[-1, 0], [0, -1.]]), np.array([[0, 0], [-1, 0.]])]
    values = ([1, 0], [0, 0], [0, 1], [1, 1])
    for variant in ('sym', 'unsym'):
        if variant == 'sym':
            ks = [np.array([[1, 0.5], [0.5, 1]]), np.eye(2), np.array([[1, 1], [1, 3.]])]
        else:
            ks = [np.array([[1, 1], [0, 1.]]), np.eye(2), np.array([[1, 0], [1, 1.]])]
        A = MPIBlockMatrix(4, 4, own, comm)
        rhs = MPIBlockVector(4, np.array(owners + [-1]), comm)
        for i in mine:
            A.set_block(i, i, coo_matrix(ks[i]))
            A.set_block(3, i, coo_matrix(border[i]))
        A.set_block(3, 3, coo_matrix(np.array([[0, 0], [0, 1.0]])))
        for i in range(4):
            A.set_row_size(i, 2)
            A.set_col_size(i, 2)
        for i in mine + [3]:
            rhs.set_block(i, np.array(values[i], dtype=np.double))
        eng = None if GPU else HostSimEngine()      # (None: HipEngine, the product's default)
        general = variant == 'unsym'
        solver = HipSchurComplementLinearSolver({i: (ScipyInterface(compute_inertia=True, engine=eng) if general else None) for i in mine},
                                                ScipyInterface(compute_inertia=True, engine=eng) if general else None,
                                                comm=comm, engine=eng)
        assert solver.do_symbolic_factorization(A).status == LinearSolverStatus.successful
        assert solver.local_block_indices == mine
        key = 'b8_%s_mpi' % variant
        for _ in range(2):
            assert solver.do_numeric_factorization(A).status == LinearSolverStatus.successful
            x = solver.do_back_solve(rhs)
            for i in mine + [3]:
                assert np.allclose(np.asarray(x.get_block(i)), golden[key + '_x'][2 * i:2 * i + 2], rtol=1e-10, atol=1e-10), (variant, i)
            assert all(x.get_block(i) is None for i in range(3) if i not in mine)
        assert np.allclose(solver.get_schur_complement(), golden[key + '_S'], rtol=1e-12, atol=1e-12)
        if not general:
            assert solver.get_inertia() == tuple(golden[key + '_inertia'])
    dist.barrier()
    dist.destroy_process_group()
    print('rank %d of %d ok' % (rank, size))


if __name__ == '__main__':
    main()
