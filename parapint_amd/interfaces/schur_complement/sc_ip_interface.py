"""Two-stage stochastic programs and time-staged dynamic problems in block-bordered KKT form, Pyomo-free.

Counterpart of ``StochasticSchurComplementInteriorPointInterface`` (parapint/interfaces/schur_complement/
sc_ip_interface.py:1028-1849) and of its MPI variant (mpi_sc_ip_interface.py:273-498) for scenarios given as
``QuadraticProgram`` objects: every scenario carries its own copy of the first-stage variables, tied to the shared
coupling variables by the nonanticipativity constraints ``L_i x_i - C_i z = 0`` (linking matrices :1287-1318).  The
KKT matrix it hands to the linear solver is exactly the reference's structure (:1245-1285, :1677-1681):

    K_i  = [[ kkt_i (4 x 4 blocks of interface.py:432-494),  [L_i 0 0 0]^T ],      border (N, i) = [ 0 | -C_i^T ]
            [ [L_i 0 0 0],                                   0 * I         ]]      corner        = 0 * I

-- nested BlockMatrix blocks, both triangles, explicit zero diagonals -- so the solver under test sees what
``ip_solve`` would give it.  With a communicator of more than one rank the scenarios are dealt round-robin
(mpi_sc_ip_interface.py:14-29) and the containers carry the ownership tables.
"""
import numpy as np
from scipy.sparse import coo_matrix, identity

from parapint_amd.interfaces.interface import InteriorPointInterface, QPInteriorPointInterface, QuadraticProgram
from parapint_amd.sparse.block_containers import (BlockMatrix, BlockVector, MPIBlockMatrix, MPIBlockVector)


def _interface_of(model):
    """The single-problem interface of one scenario / time block: a QuadraticProgram, or any object with the NLP protocol
    of interfaces/interface.py (where the reference wraps a Pyomo model, sc_ip_interface.py:156, 1166)."""
    return QPInteriorPointInterface(model) if isinstance(model, QuadraticProgram) else InteriorPointInterface(model)


class _Serial(object):
    rank, size = 0, 1

    def allreduce_sum(self, a):
        return a


def distribute_blocks(num_blocks, rank, size):
    return [ndx for ndx in range(num_blocks) if ndx % size == rank]


class StochasticSchurComplementInteriorPointInterface(object):
    """Parameters
    ----------
    scenarios: sequence of QuadraticProgram
        one subproblem per scenario (only the locally owned ones are touched)
    first_stage_indices: sequence of int arrays
        for every scenario the indices of its copies of the nonanticipative variables, in the order of the coupling
        variables (``nonanticipative_var_identifiers`` of the reference, sc_ip_interface.py:1046-1058)
    comm: communicator of parapint_amd.linalg.comm (None: serial)
    """

    def __init__(self, scenarios, first_stage_indices, comm=None):
        self._comm = _Serial() if comm is None else comm
        self._mpi = comm is not None and comm.size > 1
        self._num_scenarios = N = len(scenarios)
        if self._comm.size > N:
            raise ValueError('Cannot yet handle more processes than scenarios')     # mpi_sc_ip_interface.py:322-323
        self._ownership = {ndx: ndx % self._comm.size for ndx in range(N)}
        self._local = distribute_blocks(N, self._comm.rank, self._comm.size)
        self._num_first_stage_vars = nfs = len(first_stage_indices[0])
        self._nlps = {}
        self._linking = {}
        self._link_coupling = {}
        for ndx in self._local:
            nlp = _interface_of(scenarios[ndx])
            idx = np.asarray(first_stage_indices[ndx], dtype=np.int64)
            assert idx.size == nfs
            rows = np.arange(nfs)
            self._nlps[ndx] = nlp
            self._linking[ndx] = coo_matrix((np.ones(nfs), (rows, idx)), shape=(nfs, nlp.n_primals()))
            self._link_coupling[ndx] = coo_matrix((np.ones(nfs), (rows, rows)), shape=(nfs, nfs))
        self._primals_coupling = np.zeros(nfs)
        self._delta_coupling = np.zeros(nfs)
        self._duals_link = {ndx: np.zeros(nfs) for ndx in self._local}
        self._delta_duals_link = {ndx: np.zeros(nfs) for ndx in self._local}
        self._bounds_relaxation_factor = 0

    # ---- containers
    def _vector(self, extra_block):
        n = self._num_scenarios + (1 if extra_block else 0)
        if not self._mpi:
            return BlockVector(n)
        owner = [self._ownership[ndx] for ndx in range(self._num_scenarios)] + ([-1] if extra_block else [])
        return MPIBlockVector(n, np.asarray(owner), self._comm)

    def _per_scenario(self, fn, coupling=None):
        v = self._vector(coupling is not None)
        for ndx, nlp in self._nlps.items():
            v.set_block(ndx, fn(ndx, nlp))
        if coupling is not None:
            v.set_block(self._num_scenarios, coupling)
        return v

    @property
    def local_block_indices(self):
        return list(self._local)

    @property
    def ownership_map(self):
        return dict(self._ownership)

    def scenario_interface(self, ndx):
        return self._nlps[ndx]

    # ---- sizes
    def _sum(self, v):
        return float(self._comm.allreduce_sum(np.array([float(v)]))[0]) if self._mpi else v

    def n_primals(self):
        return int(self._sum(sum(nlp.n_primals() for nlp in self._nlps.values()))) + self._num_first_stage_vars

    def n_eq_constraints(self):
        return int(self._sum(sum(nlp.n_eq_constraints() + self._num_first_stage_vars for nlp in self._nlps.values())))

    def n_ineq_constraints(self):
        return int(self._sum(sum(nlp.n_ineq_constraints() for nlp in self._nlps.values())))

    def get_bounds_relaxation_factor(self):
        return self._bounds_relaxation_factor

    def set_bounds_relaxation_factor(self, val):
        self._bounds_relaxation_factor = val
        for nlp in self._nlps.values():
            nlp.set_bounds_relaxation_factor(val)

    def get_obj_factor(self):
        return self._nlps[self._local[0]].get_obj_factor()

    def set_obj_factor(self, obj_factor):
        for nlp in self._nlps.values():
            nlp.set_obj_factor(obj_factor)

    def set_barrier_parameter(self, barrier):
        for nlp in self._nlps.values():
            nlp.set_barrier_parameter(barrier)

    # ---- bounds and initial point (coupling variables are free, sc_ip_interface.py:1207-1225)
    def primals_lb(self):
        return self._per_scenario(lambda i, nlp: nlp.primals_lb(), np.full(self._num_first_stage_vars, -np.inf))

    def primals_ub(self):
        return self._per_scenario(lambda i, nlp: nlp.primals_ub(), np.full(self._num_first_stage_vars, np.inf))

    def ineq_lb(self):
        return self._per_scenario(lambda i, nlp: nlp.ineq_lb())

    def ineq_ub(self):
        return self._per_scenario(lambda i, nlp: nlp.ineq_ub())

    def init_primals(self):
        return self._per_scenario(lambda i, nlp: nlp.init_primals(), np.zeros(self._num_first_stage_vars))

    def init_slacks(self):
        return self._per_scenario(lambda i, nlp: nlp.init_slacks())

    def _eq_pair(self, a, b):
        sub = BlockVector(2)
        sub.set_block(0, a)
        sub.set_block(1, b)
        return sub

    def init_duals_eq(self):
        return self._per_scenario(lambda i, nlp: self._eq_pair(nlp.init_duals_eq(), np.zeros(self._num_first_stage_vars)))

    def init_duals_ineq(self):
        return self._per_scenario(lambda i, nlp: nlp.init_duals_ineq())

    def init_duals_primals_lb(self):
        return self._per_scenario(lambda i, nlp: nlp.init_duals_primals_lb(), np.zeros(self._num_first_stage_vars))

    def init_duals_primals_ub(self):
        return self._per_scenario(lambda i, nlp: nlp.init_duals_primals_ub(), np.zeros(self._num_first_stage_vars))

    def init_duals_slacks_lb(self):
        return self._per_scenario(lambda i, nlp: nlp.init_duals_slacks_lb())

    def init_duals_slacks_ub(self):
        return self._per_scenario(lambda i, nlp: nlp.init_duals_slacks_ub())

    # ---- state
    def set_primals(self, primals):
        for ndx, nlp in self._nlps.items():
            nlp.set_primals(primals.get_block(ndx))
        self._primals_coupling = np.asarray(primals.get_block(self._num_scenarios), dtype=np.double)

    def get_primals(self):
        return self._per_scenario(lambda i, nlp: nlp.get_primals(), self._primals_coupling)

    def set_slacks(self, slacks):
        for ndx, nlp in self._nlps.items():
            nlp.set_slacks(slacks.get_block(ndx))

    def get_slacks(self):
        return self._per_scenario(lambda i, nlp: nlp.get_slacks())

    def set_duals_eq(self, duals_eq):
        for ndx, nlp in self._nlps.items():
            sub = duals_eq.get_block(ndx)
            nlp.set_duals_eq(sub.get_block(0))
            self._duals_link[ndx] = np.asarray(sub.get_block(1), dtype=np.double)

    def get_duals_eq(self):
        return self._per_scenario(lambda i, nlp: self._eq_pair(nlp.get_duals_eq(), self._duals_link[i]))

    def set_duals_ineq(self, duals_ineq):
        for ndx, nlp in self._nlps.items():
            nlp.set_duals_ineq(duals_ineq.get_block(ndx))

    def get_duals_ineq(self):
        return self._per_scenario(lambda i, nlp: nlp.get_duals_ineq())

    def set_duals_primals_lb(self, duals):
        for ndx, nlp in self._nlps.items():
            nlp.set_duals_primals_lb(duals.get_block(ndx))

    def set_duals_primals_ub(self, duals):
        for ndx, nlp in self._nlps.items():
            nlp.set_duals_primals_ub(duals.get_block(ndx))

    def set_duals_slacks_lb(self, duals):
        for ndx, nlp in self._nlps.items():
            nlp.set_duals_slacks_lb(duals.get_block(ndx))

    def set_duals_slacks_ub(self, duals):
        for ndx, nlp in self._nlps.items():
            nlp.set_duals_slacks_ub(duals.get_block(ndx))

    def get_duals_primals_lb(self):
        return self._per_scenario(lambda i, nlp: nlp.get_duals_primals_lb(), np.zeros(self._num_first_stage_vars))

    def get_duals_primals_ub(self):
        return self._per_scenario(lambda i, nlp: nlp.get_duals_primals_ub(), np.zeros(self._num_first_stage_vars))

    def get_duals_slacks_lb(self):
        return self._per_scenario(lambda i, nlp: nlp.get_duals_slacks_lb())

    def get_duals_slacks_ub(self):
        return self._per_scenario(lambda i, nlp: nlp.get_duals_slacks_ub())

    # ---- evaluations
    def evaluate_objective(self):
        return self._sum(sum(nlp.evaluate_objective() for nlp in self._nlps.values()))

    def evaluate_grad_objective(self):
        return self._per_scenario(lambda i, nlp: nlp.evaluate_grad_objective(), np.zeros(self._num_first_stage_vars))

    def evaluate_eq_constraints(self):
        return self._per_scenario(lambda i, nlp: self._eq_pair(
            nlp.evaluate_eq_constraints(),
            self._linking[i] @ nlp.get_primals() - self._link_coupling[i] @ self._primals_coupling))

    def evaluate_ineq_constraints(self):
        return self._per_scenario(lambda i, nlp: nlp.evaluate_ineq_constraints())

    def grad_lag_primals_terms(self):
        """J_eq^T y_eq + J_ineq^T y_ineq of the whole problem, per block (the convergence check needs it; the
        reference multiplies the block Jacobians, interior_point.py:236-238)."""
        def per(i, nlp):
            return (nlp.evaluate_jacobian_eq().T @ nlp.get_duals_eq() + self._linking[i].T @ self._duals_link[i] +
                    nlp.evaluate_jacobian_ineq().T @ nlp.get_duals_ineq())
        last = np.zeros(self._num_first_stage_vars)
        for i in self._nlps:
            last -= self._link_coupling[i].T @ self._duals_link[i]
        if self._mpi:
            last = self._comm.allreduce_sum(last)
        return self._per_scenario(per, last)

    def _jacobian(self, rows_of, eq):
        """Block Jacobian with one block row per scenario / time block and one block column more for the coupling
        variables (sc_ip_interface.py:243-272, 1245-1270); rows_of(ndx, nlp) -> (diagonal block, coupling block)."""
        N = self._num_scenarios
        if self._mpi:
            owner = -np.ones((N, N + 1), dtype=np.int64)
            for ndx in range(N):
                owner[ndx, ndx] = owner[ndx, N] = self._ownership[ndx]
            jac = MPIBlockMatrix(N, N + 1, owner, self._comm)
        else:
            jac = BlockMatrix(N, N + 1)
        jac.set_col_size(N, self._num_first_stage_vars)
        for ndx, nlp in self._nlps.items():
            diag, coupling = rows_of(ndx, nlp)
            jac.set_block(ndx, ndx, diag)
            if coupling is not None:
                jac.set_block(ndx, N, coupling)
        return jac

    def _stack(self, mats, ncols):
        sub = BlockMatrix(len(mats), 1)
        sub.set_col_size(0, ncols)
        for k, m in enumerate(mats):
            sub.set_row_size(k, m.shape[0])
            if m.shape[0]:
                sub.set_block(k, 0, coo_matrix(m))
        return sub

    def evaluate_jacobian_eq(self):
        """Rows: per scenario [its equality constraints | nonanticipativity L x - C z] (:1245-1260)."""
        nfs = self._num_first_stage_vars

        def rows(ndx, nlp):
            me = nlp.n_eq_constraints()
            return (self._stack([nlp.evaluate_jacobian_eq(), self._linking[ndx]], nlp.n_primals()),
                    self._stack([coo_matrix((me, nfs)), -self._link_coupling[ndx]], nfs))
        return self._jacobian(rows, True)

    def evaluate_jacobian_ineq(self):
        return self._jacobian(lambda ndx, nlp: (coo_matrix(nlp.evaluate_jacobian_ineq()), None), False)

    # ---- the KKT system (sc_ip_interface.py:1245-1285, 1677-1696; mpi_...:470-478)
    def _matrix(self):
        N = self._num_scenarios
        if not self._mpi:
            return BlockMatrix(N + 1, N + 1)
        owner = -np.ones((N + 1, N + 1), dtype=np.int64)
        for ndx in range(N):
            owner[ndx, ndx] = owner[N, ndx] = owner[ndx, N] = self._ownership[ndx]
        return MPIBlockMatrix(N + 1, N + 1, owner, self._comm)

    def evaluate_primal_dual_kkt_matrix(self, timer=None, only=None):
        """only: scenario indices to evaluate (default: all local ones) -- for callers that need the blocks of a few
        scenarios (the device producer reads the common pattern off scenario 0)."""
        N, nfs = self._num_scenarios, self._num_first_stage_vars
        kkt = self._matrix()
        for ndx, nlp in self._nlps.items():
            if only is not None and ndx not in only:
                continue
            n, me, mi = nlp.n_primals(), nlp.n_eq_constraints(), nlp.n_ineq_constraints()
            sub = BlockMatrix(2, 2)
            sub.set_block(0, 0, nlp.evaluate_primal_dual_kkt_matrix())
            row_1 = BlockMatrix(1, 4)
            row_1.set_row_size(0, nfs)
            for j, size in enumerate((n, mi, me, mi)):
                row_1.set_col_size(j, size)
            row_1.set_block(0, 0, self._linking[ndx])
            sub.set_block(1, 0, row_1)
            sub.set_block(0, 1, row_1.transpose())
            ptb = identity(nfs, format='coo')
            ptb.data.fill(0)
            sub.set_block(1, 1, ptb)
            kkt.set_block(ndx, ndx, sub)
            border = BlockMatrix(1, 2)
            border.set_row_size(0, nfs)
            border.set_col_size(0, n + me + 2 * mi)
            border.set_block(0, 1, (-self._link_coupling[ndx].transpose()).tocoo())
            kkt.set_block(N, ndx, border)
            kkt.set_block(ndx, N, border.transpose())
        ptb = identity(nfs, format='coo')
        ptb.data.fill(0)
        kkt.set_block(N, N, ptb)
        if self._mpi:
            for ndx in range(N):
                if ndx not in self._nlps:
                    pass                      # sizes of remote blocks are not needed by the solver (it only visits its own)
        return kkt

    def evaluate_primal_dual_kkt_rhs(self, timer=None):
        N = self._num_scenarios
        rhs = self._vector(True)
        last = np.zeros(self._num_first_stage_vars)
        for ndx, nlp in self._nlps.items():
            sub_rhs = nlp.evaluate_primal_dual_kkt_rhs()
            sub_rhs.set_block(0, sub_rhs.get_block(0) - self._linking[ndx].T @ self._duals_link[ndx])
            pair = BlockVector(2)
            pair.set_block(0, sub_rhs)
            pair.set_block(1, self._link_coupling[ndx] @ self._primals_coupling - self._linking[ndx] @ nlp.get_primals())
            rhs.set_block(ndx, pair)
            last += self._link_coupling[ndx].T @ self._duals_link[ndx]
        if self._mpi:
            last = self._comm.allreduce_sum(last)
        rhs.set_block(N, last)
        return rhs

    def set_primal_dual_kkt_solution(self, sol):
        for ndx, nlp in self._nlps.items():
            blk = sol.get_block(ndx)
            if not hasattr(blk, 'get_block'):           # a flat block from a solver that does not keep the nesting
                flat = np.asarray(blk)
                n, me, mi = nlp.n_primals(), nlp.n_eq_constraints(), nlp.n_ineq_constraints()
                inner = BlockVector(4)
                off = 0
                for j, size in enumerate((n, mi, me, mi)):
                    inner.set_block(j, flat[off:off + size])
                    off += size
                link = flat[off:]
            else:
                inner, link = blk.get_block(0), blk.get_block(1)
            nlp.set_primal_dual_kkt_solution(inner)
            self._delta_duals_link[ndx] = np.asarray(link.flatten() if hasattr(link, 'get_block') else link)
        self._delta_coupling = np.asarray(sol.get_block(self._num_scenarios), dtype=np.double)

    def get_delta_primals(self):
        return self._per_scenario(lambda i, nlp: nlp.get_delta_primals(), self._delta_coupling)

    def get_delta_slacks(self):
        return self._per_scenario(lambda i, nlp: nlp.get_delta_slacks())

    def get_delta_duals_eq(self):
        return self._per_scenario(lambda i, nlp: self._eq_pair(nlp.get_delta_duals_eq(), self._delta_duals_link[i]))

    def get_delta_duals_ineq(self):
        return self._per_scenario(lambda i, nlp: nlp.get_delta_duals_ineq())

    def get_delta_duals_primals_lb(self):
        return self._per_scenario(lambda i, nlp: nlp.get_delta_duals_primals_lb(), np.zeros(self._num_first_stage_vars))

    def get_delta_duals_primals_ub(self):
        return self._per_scenario(lambda i, nlp: nlp.get_delta_duals_primals_ub(), np.zeros(self._num_first_stage_vars))

    def get_delta_duals_slacks_lb(self):
        return self._per_scenario(lambda i, nlp: nlp.get_delta_duals_slacks_lb())

    def get_delta_duals_slacks_ub(self):
        return self._per_scenario(lambda i, nlp: nlp.get_delta_duals_slacks_ub())

    # ---- inertia correction (sc_ip_interface.py:1736-1757)
    def regularize_equality_gradient(self, kkt, coef, copy_kkt=True):
        if copy_kkt:
            kkt = kkt.copy()
        for ndx, nlp in self._nlps.items():
            nlp.regularize_equality_gradient(kkt=kkt.get_block(ndx, ndx).get_block(0, 0), coef=coef, copy_kkt=False)
            kkt.get_block(ndx, ndx).set_block(1, 1, (coef * identity(self._num_first_stage_vars, format='coo')).tocoo())
        return kkt

    def regularize_hessian(self, kkt, coef, copy_kkt=True):
        if copy_kkt:
            kkt = kkt.copy()
        for ndx, nlp in self._nlps.items():
            nlp.regularize_hessian(kkt=kkt.get_block(ndx, ndx).get_block(0, 0), coef=coef, copy_kkt=False)
        N = self._num_scenarios
        kkt.set_block(N, N, (coef * identity(self._num_first_stage_vars, format='coo')).tocoo())
        return kkt


class DynamicSchurComplementInteriorPointInterface(StochasticSchurComplementInteriorPointInterface):
    """Dynamic optimisation problems cut into time blocks, Pyomo-free: counterpart of
    ``DynamicSchurComplementInteriorPointInterface`` (sc_ip_interface.py:13-1026) and of its MPI variant
    (mpi_sc_ip_interface.py:32-270).  Derive from it and implement ``build_model_for_time_block``; call this ``__init__``
    at the END of the derived ``__init__`` (it builds the time blocks), as the reference asks.

    The coupling variables are the states between two time blocks (n_states * (T - 1) of them, free).  Block t ties its
    start states to the coupling states z_{t-1} (backward link, absent for t = 0; its multipliers live IN the block) and
    its end states to z_t (forward link, absent for t = T - 1; its multipliers live in the COUPLING block), so the KKT
    system is (sc_ip_interface.py:274-357)

        K_t = [[ kkt_t, [Lb_t 0 0 0]^T ], [ [Lb_t 0 0 0], 0 * I ]]
        border (T, t) = [[ [Lf_t 0 0 0] on the rows of block t's forward multipliers, 0 ], [ 0, -Lbc_t^T ]]
        corner = [[ 0 * I, -Lfc ], [ -Lfc^T, 0 * I ]]       rows: forward multipliers of blocks 0..T-2, then z

    and S is block-banded: a time block touches only the coupling rows of its own two links.

    Parameters
    ----------
    start_t, end_t: float
        the time horizon; block ndx covers [delta * ndx, delta * (ndx + 1)], delta = (end_t - start_t) / num_time_blocks
    num_time_blocks: int
    comm: communicator of parapint_amd.linalg.comm (None: serial); block ndx belongs to rank ndx % size
    """

    def __init__(self, start_t, end_t, num_time_blocks, comm=None):
        self._comm = _Serial() if comm is None else comm
        self._mpi = comm is not None and comm.size > 1
        self._num_time_blocks = self._num_scenarios = T = int(num_time_blocks)
        if self._comm.size > T:
            raise ValueError('Cannot yet handle more processes than time blocks')   # mpi_sc_ip_interface.py:79-80
        self._ownership = {ndx: ndx % self._comm.size for ndx in range(T)}
        self._local = distribute_blocks(T, self._comm.rank, self._comm.size)
        self._num_states = None
        self._nlps, self._link_backward, self._link_forward = {}, {}, {}
        delta_t = (end_t - start_t) / T
        for ndx in self._local:                                   # mpi_sc_ip_interface.py:178-215
            qp, start_states, end_states = self.build_model_for_time_block(
                ndx=ndx, start_t=delta_t * ndx, end_t=delta_t * (ndx + 1), add_init_conditions=(ndx == 0))
            self._nlps[ndx] = nlp = _interface_of(qp)
            assert len(start_states) == len(end_states)
            if self._num_states is not None:
                assert self._num_states == len(start_states)
            else:
                self._num_states = len(start_states)
            ns, n = self._num_states, nlp.n_primals()
            b_idx = np.asarray(start_states if ndx != 0 else [], dtype=np.int64)
            f_idx = np.asarray(end_states if ndx != T - 1 else [], dtype=np.int64)
            self._link_backward[ndx] = coo_matrix((np.ones(b_idx.size), (np.arange(b_idx.size), b_idx)), shape=(b_idx.size, n))
            self._link_forward[ndx] = coo_matrix((np.ones(f_idx.size), (np.arange(f_idx.size), f_idx)), shape=(f_idx.size, n))
        ns = self._num_states
        self._num_first_stage_vars = ncz = ns * (T - 1)           # (the base class sizes the coupling vectors with it)
        # coupling-side link matrices of EVERY block (:388-418, 449-478): backward -> z_{ndx-1}, forward -> z_ndx
        self._link_backward_coupling, self._link_forward_coupling = {}, {}
        k = np.arange(ns)
        for ndx in range(T):
            nb, nf = (ns if ndx != 0 else 0), (ns if ndx != T - 1 else 0)
            self._link_backward_coupling[ndx] = coo_matrix((np.ones(nb), (k[:nb], ns * (ndx - 1) + k[:nb])), shape=(nb, ncz))
            self._link_forward_coupling[ndx] = coo_matrix((np.ones(nf), (k[:nf], ns * ndx + k[:nf])), shape=(nf, ncz))
        self._primals_coupling = np.zeros(ncz)
        self._delta_coupling = np.zeros(ncz)
        self._duals_backward = {ndx: np.zeros(self._link_backward[ndx].shape[0]) for ndx in self._local}
        self._duals_forward = {ndx: np.zeros(self._link_forward[ndx].shape[0]) for ndx in self._local}
        self._delta_duals_backward = {ndx: np.zeros(self._link_backward[ndx].shape[0]) for ndx in self._local}
        self._delta_duals_forward = {ndx: np.zeros(self._link_forward[ndx].shape[0]) for ndx in self._local}
        self._bounds_relaxation_factor = 0

    def build_model_for_time_block(self, ndx, start_t, end_t, add_init_conditions):
        """Return (QuadraticProgram of the time interval, indices of its primal variables that are the states at
        start_t, indices of the states at end_t), the two lists in the same order for every time block;
        add_init_conditions is True for time block 0 only (sc_ip_interface.py:107-143)."""
        raise NotImplementedError('derived classes implement build_model_for_time_block')

    @property
    def num_states(self):
        return self._num_states

    def n_eq_constraints(self):
        return int(self._sum(sum(nlp.n_eq_constraints() for nlp in self._nlps.values()))) + 2 * self._num_first_stage_vars

    # ---- equality duals and residuals: per block [the block's own | backward link | forward link] (:184-199, 716-739)
    def _eq_triple(self, a, b, c):
        sub = BlockVector(3)
        sub.set_block(0, a)
        sub.set_block(1, b)
        sub.set_block(2, c)
        return sub

    def init_duals_eq(self):
        return self._per_scenario(lambda i, nlp: self._eq_triple(
            nlp.init_duals_eq(), np.zeros(self._link_backward[i].shape[0]), np.zeros(self._link_forward[i].shape[0])))

    def set_duals_eq(self, duals_eq):
        for ndx, nlp in self._nlps.items():
            sub = duals_eq.get_block(ndx)
            nlp.set_duals_eq(sub.get_block(0))
            self._duals_backward[ndx] = np.asarray(sub.get_block(1), dtype=np.double)
            self._duals_forward[ndx] = np.asarray(sub.get_block(2), dtype=np.double)

    def get_duals_eq(self):
        return self._per_scenario(lambda i, nlp: self._eq_triple(nlp.get_duals_eq(), self._duals_backward[i],
                                                                 self._duals_forward[i]))

    def evaluate_eq_constraints(self):
        z = self._primals_coupling
        return self._per_scenario(lambda i, nlp: self._eq_triple(
            nlp.evaluate_eq_constraints(),
            self._link_backward[i] @ nlp.get_primals() - self._link_backward_coupling[i] @ z,
            self._link_forward[i] @ nlp.get_primals() - self._link_forward_coupling[i] @ z))

    def _coupling_duals_term(self):
        last = np.zeros(self._num_first_stage_vars)
        for i in self._nlps:
            last += self._link_backward_coupling[i].T @ self._duals_backward[i]
            last += self._link_forward_coupling[i].T @ self._duals_forward[i]
        return self._comm.allreduce_sum(last) if self._mpi else last

    def grad_lag_primals_terms(self):
        def per(i, nlp):
            return (nlp.evaluate_jacobian_eq().T @ nlp.get_duals_eq() + self._link_backward[i].T @ self._duals_backward[i] +
                    self._link_forward[i].T @ self._duals_forward[i] + nlp.evaluate_jacobian_ineq().T @ nlp.get_duals_ineq())
        return self._per_scenario(per, -self._coupling_duals_term())

    def evaluate_jacobian_eq(self):
        """Rows: per time block [its equality constraints | backward link Lb x - Lbc z | forward link Lf x - Lfc z]
        (:243-272, 753-765)."""
        ncz = self._num_first_stage_vars

        def rows(ndx, nlp):
            me = nlp.n_eq_constraints()
            return (self._stack([nlp.evaluate_jacobian_eq(), self._link_backward[ndx], self._link_forward[ndx]], nlp.n_primals()),
                    self._stack([coo_matrix((me, ncz)), -self._link_backward_coupling[ndx], -self._link_forward_coupling[ndx]], ncz))
        return self._jacobian(rows, True)

    # ---- the KKT system (:274-357, 839-862; mpi_...:242-250)
    def _forward_rows(self, ndx):
        """Rows of block ndx's forward multipliers inside the coupling block."""
        return self._num_states * ndx + np.arange(self._link_forward[ndx].shape[0])

    def evaluate_primal_dual_kkt_matrix(self, timer=None, only=None):
        T, ns, ncz = self._num_time_blocks, self._num_states, self._num_first_stage_vars
        kkt = self._matrix()
        for ndx, nlp in self._nlps.items():
            if only is not None and ndx not in only:
                continue
            n, me, mi = nlp.n_primals(), nlp.n_eq_constraints(), nlp.n_ineq_constraints()
            nb = self._link_backward[ndx].shape[0]
            sub = BlockMatrix(2, 2)
            sub.set_block(0, 0, nlp.evaluate_primal_dual_kkt_matrix())
            row_1 = BlockMatrix(1, 4)
            row_1.set_row_size(0, nb)
            for j, size in enumerate((n, mi, me, mi)):
                row_1.set_col_size(j, size)
            if nb:
                row_1.set_block(0, 0, self._link_backward[ndx])
            sub.set_block(1, 0, row_1)
            sub.set_block(0, 1, row_1.transpose())
            ptb = identity(nb, format='coo')
            ptb.data.fill(0)
            sub.set_block(1, 1, ptb)
            kkt.set_block(ndx, ndx, sub)
            # border: the forward link on the rows of this block's forward multipliers, -Lbc^T on the rows of z
            border = BlockMatrix(2, 2)
            border.set_row_size(0, ncz)
            border.set_row_size(1, ncz)
            border.set_col_size(0, n + me + 2 * mi)
            border.set_col_size(1, nb)
            Lf = self._link_forward[ndx]
            if Lf.shape[0]:
                border.set_block(0, 0, coo_matrix((Lf.data, (self._forward_rows(ndx)[Lf.row], Lf.col)),
                                                  shape=(ncz, n + me + 2 * mi)))
            if nb:
                border.set_block(1, 1, (-self._link_backward_coupling[ndx].transpose()).tocoo())
            kkt.set_block(T, ndx, border)
            kkt.set_block(ndx, T, border.transpose())
        kkt.set_block(T, T, self._corner(0.0, 0.0))
        return kkt

    def _corner(self, c_eq, c_hess):
        """[[c_eq * I, -Lfc], [-Lfc^T, c_hess * I]]: explicit diagonals (zero unless regularised, :340-356, 914-933)."""
        ncz = self._num_first_stage_vars
        i = np.arange(ncz)
        block = BlockMatrix(2, 2)
        block.set_block(0, 0, coo_matrix((np.full(ncz, float(c_eq)), (i, i)), shape=(ncz, ncz)))
        block.set_block(1, 1, coo_matrix((np.full(ncz, float(c_hess)), (i, i)), shape=(ncz, ncz)))
        # stacked forward coupling matrices of blocks 0..T-2: the identity
        block.set_block(1, 0, coo_matrix((-np.ones(ncz), (i, i)), shape=(ncz, ncz)))
        block.set_block(0, 1, coo_matrix((-np.ones(ncz), (i, i)), shape=(ncz, ncz)))
        return block

    def evaluate_primal_dual_kkt_rhs(self, timer=None):
        T, ncz = self._num_time_blocks, self._num_first_stage_vars
        z = self._primals_coupling
        rhs = self._vector(True)
        forward = np.zeros(ncz)
        for ndx, nlp in self._nlps.items():
            sub_rhs = nlp.evaluate_primal_dual_kkt_rhs()
            sub_rhs.set_block(0, sub_rhs.get_block(0) - (self._link_backward[ndx].T @ self._duals_backward[ndx] +
                                                         self._link_forward[ndx].T @ self._duals_forward[ndx]))
            pair = BlockVector(2)
            pair.set_block(0, sub_rhs)
            pair.set_block(1, self._link_backward_coupling[ndx] @ z - self._link_backward[ndx] @ nlp.get_primals())
            rhs.set_block(ndx, pair)
            forward[self._forward_rows(ndx)] = self._link_forward_coupling[ndx] @ z - self._link_forward[ndx] @ nlp.get_primals()
        if self._mpi:                                             # (the blocks of the other ranks are zero)
            forward = self._comm.allreduce_sum(forward)
        last = BlockVector(2)
        last.set_block(0, forward)
        last.set_block(1, self._coupling_duals_term())
        rhs.set_block(T, last)
        return rhs

    def set_primal_dual_kkt_solution(self, sol):
        T, ncz = self._num_time_blocks, self._num_first_stage_vars
        last = sol.get_block(T)
        last = np.asarray(last.flatten() if hasattr(last, 'get_block') else last, dtype=np.double)
        for ndx, nlp in self._nlps.items():
            blk = sol.get_block(ndx)
            flat = np.asarray(blk.flatten() if hasattr(blk, 'get_block') else blk, dtype=np.double)
            n, me, mi = nlp.n_primals(), nlp.n_eq_constraints(), nlp.n_ineq_constraints()
            inner = BlockVector(4)
            off = 0
            for j, size in enumerate((n, mi, me, mi)):
                inner.set_block(j, flat[off:off + size])
                off += size
            nlp.set_primal_dual_kkt_solution(inner)
            self._delta_duals_backward[ndx] = flat[off:].copy()
            self._delta_duals_forward[ndx] = last[self._forward_rows(ndx)].copy()
        self._delta_coupling = last[ncz:].copy()

    def get_delta_duals_eq(self):
        return self._per_scenario(lambda i, nlp: self._eq_triple(nlp.get_delta_duals_eq(), self._delta_duals_backward[i],
                                                                 self._delta_duals_forward[i]))

    # ---- inertia correction (:903-933)
    def _corner_coefs(self, kkt):
        block = kkt.get_block(self._num_time_blocks, self._num_time_blocks)
        d0, d1 = block.get_block(0, 0).tocoo().data, block.get_block(1, 1).tocoo().data
        return (float(d0[0]) if d0.size else 0.0), (float(d1[0]) if d1.size else 0.0)

    def regularize_equality_gradient(self, kkt, coef, copy_kkt=True):
        if copy_kkt:
            kkt = kkt.copy()
        for ndx, nlp in self._nlps.items():
            nlp.regularize_equality_gradient(kkt=kkt.get_block(ndx, ndx).get_block(0, 0), coef=coef, copy_kkt=False)
            nb = self._link_backward[ndx].shape[0]
            kkt.get_block(ndx, ndx).set_block(1, 1, (coef * identity(nb, format='coo')).tocoo())
        T = self._num_time_blocks
        kkt.set_block(T, T, self._corner(coef, self._corner_coefs(kkt)[1]))
        return kkt

    def regularize_hessian(self, kkt, coef, copy_kkt=True):
        if copy_kkt:
            kkt = kkt.copy()
        for ndx, nlp in self._nlps.items():
            nlp.regularize_hessian(kkt=kkt.get_block(ndx, ndx).get_block(0, 0), coef=coef, copy_kkt=False)
        T = self._num_time_blocks
        kkt.set_block(T, T, self._corner(self._corner_coefs(kkt)[0], coef))
        return kkt


# the reference keeps the serial and the MPI flavour in two classes; here the communicator decides
MPIStochasticSchurComplementInteriorPointInterface = StochasticSchurComplementInteriorPointInterface
MPIDynamicSchurComplementInteriorPointInterface = DynamicSchurComplementInteriorPointInterface
