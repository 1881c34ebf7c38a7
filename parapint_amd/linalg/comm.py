"""Communicator abstraction for the one data-parallel exchange of the path.

The reference freezes ``MPI.COMM_WORLD`` at import time
(``mpi_explicit_schur_complement.py:14-16``, quirk Q1).  Here the communicator is
injected at construction.  Three implementations:

  * ``SerialComm``   -- size 1, every collective is the identity.
  * ``TorchComm``    -- ``torch.distributed`` process group; backend ``nccl`` (= RCCL
                        over xGMI on ROCm) for device tensors, ``gloo`` for the CPU
                        rehearsal of the multi-rank host logic.
  * ``MPI4PyComm``   -- adapter used only when mpi4py is importable (it is not in the
                        build image) so the class can drop into an mpirun-launched
                        parapint unchanged.

Collectives needed: all-reduce(sum) of the Schur buffer
(+ packed status / inertia words) once per numeric factorisation
(reference ``:343``, ``:21``, ``:427-429``), all-reduce(sum) of the coupling rhs once
per back-solve (``:387``); for the interior-point step on device-resident iterates two
all-gathers of a handful of scalars per iteration (step lengths; convergence measures and
the coupling block of the right-hand side, ``mpi_sc_ip_interface.py:470-478``).
"""
import numpy as np


class SerialComm(object):
    rank = 0
    size = 1
    device_collectives = False

    def allreduce_sum(self, arr):
        return arr

    def allreduce_max(self, arr):
        return arr

    allreduce_max_int = allreduce_max

    def allreduce_sum_tensor_(self, tensor):
        return tensor

    def allgather(self, arr):
        return np.asarray(arr)[None]

    def barrier(self):
        pass


class TorchComm(object):
    """torch.distributed-backed communicator (nccl == RCCL on ROCm, or gloo)."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self._torch = torch
        self._dist = dist
        self._group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.device_collectives = (self.backend == 'nccl')
        # issue the data-path collectives even in a one-rank group (lets a one-GPU box exercise the RCCL calls on
        # the solver's device buffers; a sum over one rank is the identity)
        self.always_reduce = False

    def _host_allreduce(self, arr, op):
        torch = self._torch
        a = np.ascontiguousarray(arr)
        t = torch.from_numpy(a.copy())
        if self.device_collectives:
            t = t.cuda()
        self._dist.all_reduce(t, op=op, group=self._group)
        return t.cpu().numpy()

    def allreduce_sum(self, arr):
        return self._host_allreduce(arr, self._dist.ReduceOp.SUM)

    def allreduce_max(self, arr):
        return self._host_allreduce(np.asarray(arr), self._dist.ReduceOp.MAX)

    allreduce_max_int = allreduce_max

    def allreduce_sum_tensor_(self, tensor):
        """In-place all-reduce of a (device) torch tensor: the hot collective."""
        self._dist.all_reduce(tensor, op=self._dist.ReduceOp.SUM, group=self._group)
        return tensor

    def allgather(self, arr):
        """[size][...] host array of every rank's `arr` (the scalars of the interior-point step, rank order)."""
        torch = self._torch
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.double).copy())
        if self.device_collectives:
            t = t.cuda()
        out = [torch.empty_like(t) for _ in range(self.size)]
        self._dist.all_gather(out, t, group=self._group)
        return np.stack([o.cpu().numpy() for o in out])

    def allgather_tensor_(self, table, local):
        """table[r] = local of rank r (device tensors, RCCL)."""
        self._dist.all_gather_into_tensor(table.view(-1), local.view(-1), group=self._group)
        return table

    def barrier(self):
        self._dist.barrier(group=self._group)


class MPI4PyComm(object):
    """Adapter over an mpi4py communicator (only if mpi4py is installed)."""
    device_collectives = False

    def __init__(self, comm=None):
        from mpi4py import MPI  # noqa: deliberately lazy, mpi4py is optional
        self._MPI = MPI
        self._comm = MPI.COMM_WORLD if comm is None else comm
        self.rank = self._comm.Get_rank()
        self.size = self._comm.Get_size()

    def allreduce_sum(self, arr):
        a = np.ascontiguousarray(arr)
        out = np.zeros_like(a)
        self._comm.Allreduce(a, out)
        return out

    def allreduce_max(self, arr):
        a = np.ascontiguousarray(arr)
        out = np.zeros_like(a)
        self._comm.Allreduce(a, out, op=self._MPI.MAX)
        return out

    allreduce_max_int = allreduce_max

    def allreduce_sum_tensor_(self, tensor):
        host = tensor.cpu().numpy()
        tensor.copy_(tensor.new_tensor(self.allreduce_sum(host)))
        return tensor

    def allgather(self, arr):
        a = np.ascontiguousarray(arr, dtype=np.double)
        out = np.zeros((self.size,) + a.shape)
        self._comm.Allgather(a, out)
        return out

    def barrier(self):
        self._comm.Barrier()


def default_comm():
    """SerialComm unless torch.distributed has been initialised by the launcher."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return TorchComm()
    except Exception:
        pass
    return SerialComm()
