"""ctypes binding of libidto_hip.so (include/idto_hip.h): the MI355X implementation
of IDTO's per-iteration gradient/Hessian path.

There is no CPU fallback: loading fails loudly if the library has not been built
(`./build.sh` or `__graft_entry__.build()`), and creating a context fails if no
HIP device is present.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .model import CContactParams, CModel, CProblem, Model, dptr
from .problem import ProblemDefinition, SolverParameters

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IDTO_HIP_LIB") or os.path.join(_HERE, "libidto_hip.so")   # (IDTO_HIP_LIB: a variant build, tools/fd_variants.sh)

ARR = dict(q=0, v=1, a=2, tau=3, nplus=4, dtau_dqm=5, dtau_dqt=6, dtau_dqp=7, gradient=8, H_A=9, H_B=10, H_C=11,
           step=12, cost=13, slab=14, debug=15, hbands=16, tr_dq=17, tr_w=18, tr_scale=19, asm_terms=20, con_S=21, con_lambda=22)

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: build the HIP extension first (./build.sh). "
                "idto_amd has no CPU fallback for the hot path.")
        # torch ships its own ROCm runtime; load it FIRST so that this library binds to the
        # same libamdhip64 (two HIP runtimes in one process lose the device for one of them).
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        L.idto_hip_last_error.restype = C.c_char_p
        L.idto_hip_create.argtypes = [C.POINTER(CModel), C.POINTER(CProblem), C.POINTER(CContactParams), C.c_int,
                                      C.POINTER(C.c_void_p)]
        L.idto_hip_create_batch.argtypes = [C.POINTER(CModel), C.POINTER(CProblem), C.POINTER(CContactParams), C.c_int,
                                            C.c_int, C.POINTER(C.c_void_p)]
        L.idto_hip_batch_size.argtypes = [C.c_void_p]
        L.idto_hip_set_problem_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(CProblem)]
        L.idto_hip_set_q_batch.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.idto_hip_gn_step_batch.argtypes = [C.c_void_p]
        L.idto_hip_get_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.idto_hip_solver_status_batch.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.idto_hip_comm_unique_id.argtypes = [C.c_char_p, C.c_int]
        L.idto_hip_comm_init.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
        L.idto_hip_comm_init_all.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        L.idto_hip_gn_step_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        for f in ("comm_destroy", "allgather_slab", "gn_step_sharded"):
            getattr(L, "idto_hip_" + f).argtypes = [C.c_void_p]
        L.idto_hip_tr_prepare.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.idto_hip_tr_trial.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.POINTER(C.c_double)]
        L.idto_hip_tr_accept.argtypes = [C.c_void_p]
        L.idto_hip_tr_solve.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                        C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.idto_hip_tr_solve_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_double,
                                              C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.idto_hip_tr_solve_batch_constrained.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double),
                                                          C.c_double, C.c_double, C.POINTER(C.c_int), C.c_int,
                                                          C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.idto_hip_tr_reject.argtypes = [C.c_void_p]
        L.idto_hip_tr_set_scale_memory.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.idto_hip_tr_set_convergence.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.idto_hip_set_unactuated_dofs.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
        L.idto_hip_destroy.argtypes = [C.c_void_p]
        L.idto_hip_set_problem.argtypes = [C.c_void_p, C.POINTER(CProblem)]
        L.idto_hip_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.idto_hip_get_stream.argtypes = [C.c_void_p]
        L.idto_hip_get_stream.restype = C.c_void_p
        L.idto_hip_set_shard.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.idto_hip_set_q.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.idto_hip_set_q_device.argtypes = [C.c_void_p, C.c_void_p]
        L.idto_hip_trial_cost.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.POINTER(C.c_double)]
        for f in ("eval_tau", "eval_partials", "grad_hess", "gn_step", "sync", "timing_reset"):
            getattr(L, "idto_hip_" + f).argtypes = [C.c_void_p]
        L.idto_hip_factor_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.idto_hip_solve_host.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double)]
        L.idto_hip_solve_dense_ldlt.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.idto_hip_constraint_schur.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_double),
                                                C.POINTER(C.c_double)]
        L.idto_hip_constraint_schur_begin.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
        L.idto_hip_prefetch.argtypes = [C.c_void_p, C.c_int]
        L.idto_hip_constraint_solve.argtypes = [C.c_void_p] + [C.POINTER(C.c_double)] * 4
        L.idto_hip_constraint_step.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                               C.POINTER(C.c_double)]
        L.idto_hip_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.idto_hip_get_option.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]
        L.idto_hip_timing_enable.argtypes = [C.c_void_p, C.c_int]
        L.idto_hip_timing_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.idto_hip_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        L.idto_hip_device_ptr.argtypes = [C.c_void_p, C.c_int]
        L.idto_hip_device_ptr.restype = C.c_void_p
        L.idto_hip_array_size.argtypes = [C.c_void_p, C.c_int]
        L.idto_hip_array_size.restype = C.c_long
        L.idto_hip_slab_stride.argtypes = [C.c_void_p]
        L.idto_hip_solver_status.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.idto_hip_math_probe.argtypes = [C.c_int, C.POINTER(C.c_double), C.c_int] + [C.POINTER(C.c_double)] * 6
        _lib = L
    return _lib


EXPORTED_SYMBOLS = [
    "idto_hip_last_error", "idto_hip_create", "idto_hip_destroy", "idto_hip_set_problem", "idto_hip_set_stream",
    "idto_hip_get_stream", "idto_hip_set_shard", "idto_hip_set_q", "idto_hip_set_q_device", "idto_hip_eval_tau", "idto_hip_eval_tau_partials", "idto_hip_trial_cost", "idto_hip_constraint_schur",
    "idto_hip_constraint_schur_begin", "idto_hip_constraint_solve", "idto_hip_constraint_step", "idto_hip_prefetch",
    "idto_hip_eval_partials", "idto_hip_grad_hess", "idto_hip_factor_solve", "idto_hip_gn_step", "idto_hip_solve_host",
    "idto_hip_solve_dense_ldlt", "idto_hip_dense_solve_count",
    "idto_hip_set_option", "idto_hip_get_option",
    "idto_hip_timing_enable", "idto_hip_timing_reset", "idto_hip_timing_get", "idto_hip_sync", "idto_hip_get",
    "idto_hip_device_ptr", "idto_hip_array_size", "idto_hip_slab_stride", "idto_hip_math_probe",
    "idto_hip_solver_status", "idto_hip_create_batch", "idto_hip_batch_size", "idto_hip_set_problem_batch",
    "idto_hip_set_q_batch", "idto_hip_gn_step_batch", "idto_hip_get_batch", "idto_hip_get_many", "idto_hip_solver_status_batch",
    "idto_hip_tr_prepare", "idto_hip_tr_trial", "idto_hip_tr_accept", "idto_hip_tr_reject", "idto_hip_tr_set_scale_memory", "idto_hip_tr_set_convergence", "idto_hip_tr_solve", "idto_hip_tr_solve_fetch", "idto_hip_tr_solve_batch", "idto_hip_tr_solve_batch_constrained", "idto_hip_set_unactuated_dofs",
    "idto_hip_rccl_info", "idto_hip_comm_unique_id", "idto_hip_comm_init", "idto_hip_comm_init_all", "idto_hip_comm_destroy",
    "idto_hip_allgather_slab", "idto_hip_gn_step_sharded", "idto_hip_gn_step_multi", "idto_hip_eval_partials_multi",
    "idto_hip_trace_enable", "idto_hip_trace_mark", "idto_hip_trace_dump",
]


class HipError(RuntimeError):
    pass


class FactorizationFailed(HipError):
    """IDTO_HIP_FACTORIZATION_FAILED: H is not numerically positive definite (the reference's
    PentaDiagonalFactorizationStatus::kFailure, optimizer/penta_diagonal_solver.h:181-185)"""


class SolverTimeout(HipError):
    """IDTO_HIP_SOLVER_TIMEOUT: a multi-workgroup solver launch did not find its partner workgroups resident; the
    context has stepped down to a variant with fewer co-resident workgroups - repeat the call"""


FACTORIZATION_FAILED = 2
SOLVER_TIMEOUT = 3


def _chk(rc):
    if rc == FACTORIZATION_FAILED:
        raise FactorizationFailed(lib().idto_hip_last_error().decode())
    if rc == SOLVER_TIMEOUT:
        raise SolverTimeout(lib().idto_hip_last_error().decode())
    if rc != 0:
        raise HipError(f"idto_hip error {rc}: {lib().idto_hip_last_error().decode()}")


class HipPath:
    """One problem resident on one MI355X: the device side of TrajectoryOptimizer."""

    def __init__(self, model: Model, prob, params: SolverParameters, device: int = 0):
        """`prob`: one ProblemDefinition, or a list of them = a batch of problems of the same model and
        horizon advanced together (idto_hip_create_batch); see HipBatch"""
        L = lib()
        probs = list(prob) if isinstance(prob, (list, tuple)) else [prob]
        self.batch = len(probs)
        self.model, self.prob, self.params = model, probs[0], params
        self.nq, self.nv, self.N = model.nq, model.nv, probs[0].num_steps
        cm, self._k1 = model.to_c()
        carr = (CProblem * self.batch)()
        self._k2 = []
        for b, pr in enumerate(probs):
            cp, keep = pr.to_c()
            carr[b] = cp
            self._k2.append(keep)
        cc = params.contact_to_c()
        h = C.c_void_p()
        _chk(L.idto_hip_create_batch(C.byref(cm), carr, C.byref(cc), int(device), self.batch, C.byref(h)))
        self.h = h
        self.device = device
        from .problem import GRADIENTS
        gm = GRADIENTS[params.gradients_method]
        if gm:  # 1 central, 2 central4 (SolverParameters::gradients_method); others are refused by the library
            _chk(L.idto_hip_set_option(self.h, b"gradients_method", gm))

    def close(self):
        if getattr(self, "h", None):
            lib().idto_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- inputs
    def set_q(self, q):
        q = np.ascontiguousarray(np.asarray(q, dtype=np.float64))
        assert q.size == (self.N + 1) * self.nq
        _chk(lib().idto_hip_set_q(self.h, dptr(q)))

    def set_q_batch(self, q):
        """q: [batch, N+1, nq]"""
        q = np.ascontiguousarray(np.asarray(q, dtype=np.float64))
        assert q.size == self.batch * (self.N + 1) * self.nq
        _chk(lib().idto_hip_set_q_batch(self.h, dptr(q)))

    def set_problem_batch(self, b: int, prob: ProblemDefinition):
        cp, keep = prob.to_c()
        _chk(lib().idto_hip_set_problem_batch(self.h, int(b), C.byref(cp)))

    def solver_status_batch(self):
        out = (C.c_int * self.batch)()
        _chk(lib().idto_hip_solver_status_batch(self.h, out))
        return [bool(x) for x in out]

    def set_q_device(self, ptr: int):
        _chk(lib().idto_hip_set_q_device(self.h, C.c_void_p(ptr)))

    def set_problem(self, prob: ProblemDefinition):
        cp, self._k2 = prob.to_c()
        self.prob = prob
        _chk(lib().idto_hip_set_problem(self.h, C.byref(cp)))

    def set_stream(self, stream_ptr: int):
        _chk(lib().idto_hip_set_stream(self.h, C.c_void_p(stream_ptr)))

    # ---- multi-GPU: RCCL communicator inside the library (include/idto_hip.h)
    def comm_init(self, unique_id: bytes, rank: int, world: int):
        """collective over the `world` ranks (ncclCommInitRank); sets this context's k-range shard"""
        _chk(lib().idto_hip_comm_init(self.h, unique_id, int(rank), int(world)))

    def comm_destroy(self):
        _chk(lib().idto_hip_comm_destroy(self.h))

    def allgather_slab(self):
        _chk(lib().idto_hip_allgather_slab(self.h))

    def gn_step_sharded(self):
        _chk(lib().idto_hip_gn_step_sharded(self.h))

    # ---- trust-region bookkeeping on the device (include/idto_hip.h idto_hip_tr_*)
    def tr_prepare(self, scaling_method: int = -1, with_lambda: bool = False):
        out = np.zeros(9)
        _chk(lib().idto_hip_tr_prepare(self.h, int(scaling_method), int(with_lambda), dptr(out)))
        return out

    def tr_trial(self, a: float, b: float, scaling: bool, normalize_quaternions: bool = False, with_lambda: bool = False,
                 speculate_scaling_method: int = -2):
        out = np.zeros(4)
        _chk(lib().idto_hip_tr_trial(self.h, float(a), float(b), int(scaling), int(normalize_quaternions),
                                     int(with_lambda), int(speculate_scaling_method), dptr(out)))
        return out

    def tr_reject(self):
        _chk(lib().idto_hip_tr_reject(self.h))

    def tr_set_scale_memory(self, D_prev=None):
        """the adaptive scalings' previous D (None: ones, a fresh state)"""
        if D_prev is None:
            _chk(lib().idto_hip_tr_set_scale_memory(self.h, None))
        else:
            _chk(lib().idto_hip_tr_set_scale_memory(self.h, dptr(np.ascontiguousarray(D_prev, dtype=np.float64))))

    def tr_solve(self, iterations: int, scaling_method: int, scaling: bool, normalize_quaternions: bool, Delta0: float,
                 Delta_max: float, eta: float = 0.0, constrained_dofs=()):
        """the whole trust-region loop on the device (idto_hip_tr_solve): (rows [iterations, 17], final Delta)"""
        rows = np.zeros((int(iterations), 17))
        delta = C.c_double(0.0)
        dofs = np.ascontiguousarray(np.asarray(constrained_dofs, dtype=np.int32))
        self.last_tr_rows = rows   # (filled even when the call reports a failed factorisation)
        _chk(lib().idto_hip_tr_solve(self.h, int(iterations), int(scaling_method), int(scaling), int(normalize_quaternions),
                                     float(Delta0), float(Delta_max), float(eta),
                                     dofs.ctypes.data_as(C.POINTER(C.c_int)) if dofs.size else None, int(dofs.size),
                                     dptr(rows), C.byref(delta)))
        return rows, delta.value

    def tr_solve_batch(self, iterations: int, scaling_method: int, scaling: bool, normalize_quaternions: bool, Delta0,
                       Delta_max: float, eta: float = 0.0):
        """idto_hip_tr_solve_batch: (rows [batch, iterations, 17], final Delta [batch])"""
        B = self.batch
        rows = np.zeros((B, int(iterations), 17))
        d0 = np.ascontiguousarray(np.broadcast_to(np.asarray(Delta0, dtype=np.float64), (B,)))
        delta = np.zeros(B)
        self.last_tr_rows = rows
        _chk(lib().idto_hip_tr_solve_batch(self.h, int(iterations), int(scaling_method), int(scaling), int(normalize_quaternions),
                                           dptr(d0), float(Delta_max), float(eta), dptr(rows), dptr(delta)))
        return rows, delta

    def tr_solve_batch_constrained(self, iterations: int, scaling_method: int, scaling: bool, normalize_quaternions: bool, Delta0,
                                   Delta_max: float, constrained_dofs, eta: float = 0.0):
        """idto_hip_tr_solve_batch_constrained: the batch's loop with the equality constraints enforced on `constrained_dofs`"""
        B = self.batch
        rows = np.zeros((B, int(iterations), 17))
        d0 = np.ascontiguousarray(np.broadcast_to(np.asarray(Delta0, dtype=np.float64), (B,)))
        delta = np.zeros(B)
        dofs = np.ascontiguousarray(np.asarray(constrained_dofs, dtype=np.int32))
        self.last_tr_rows = rows
        _chk(lib().idto_hip_tr_solve_batch_constrained(self.h, int(iterations), int(scaling_method), int(scaling),
                                                       int(normalize_quaternions), dptr(d0), float(Delta_max), float(eta),
                                                       dofs.ctypes.data_as(C.POINTER(C.c_int)), len(dofs), dptr(rows), dptr(delta)))
        return rows, delta

    def tr_set_convergence(self, tolerances=None):
        """[rel_cost, abs_cost, rel_gradient_along_dq, abs_gradient_along_dq, rel_state, abs_state] or None (no checks)"""
        if tolerances is None:
            _chk(lib().idto_hip_tr_set_convergence(self.h, None))
        else:
            t = np.ascontiguousarray(np.asarray(tolerances, dtype=np.float64))
            assert t.size == 6
            _chk(lib().idto_hip_tr_set_convergence(self.h, dptr(t)))

    def set_unactuated_dofs(self, dofs):
        dofs = np.ascontiguousarray(np.asarray(dofs, dtype=np.int32))
        _chk(lib().idto_hip_set_unactuated_dofs(self.h, dofs.ctypes.data_as(C.POINTER(C.c_int)), int(dofs.size)))

    def tr_accept(self):
        _chk(lib().idto_hip_tr_accept(self.h))

    def set_shard(self, k_begin: int, k_end: int):
        _chk(lib().idto_hip_set_shard(self.h, int(k_begin), int(k_end)))

    # ---- path pieces (asynchronous on the context's stream)
    def eval_tau(self):
        _chk(lib().idto_hip_eval_tau(self.h))

    def eval_tau_partials(self):
        """eval_tau for a q whose partials come next: one finite-difference launch for both (the next eval_partials is done)"""
        _chk(lib().idto_hip_eval_tau_partials(self.h))

    def constraint_schur(self, dofs):
        """S = J H^-1 J^T (n_eq x n_eq) and J H^-1 g for the constraint tau_t[dofs] = 0 (after grad_hess)"""
        dofs = np.ascontiguousarray(np.asarray(dofs, dtype=np.int32))
        neq = dofs.size * self.N
        S, Jy = np.empty((neq, neq), order="F"), np.empty(neq)
        _chk(lib().idto_hip_constraint_schur(self.h, dofs.ctypes.data_as(C.POINTER(C.c_int)), int(dofs.size),
                                             S.ctypes.data_as(C.POINTER(C.c_double)), dptr(Jy)))
        return S, Jy

    def constraint_solve(self, dofs, h):
        """multipliers entirely on the device: returns (ok, lambda, H^-1 (g + J^T lambda), J^T lambda);
        ok False: S numerically singular, use constraint_schur + a pivoted host solve"""
        dofs = np.ascontiguousarray(np.asarray(dofs, dtype=np.int32))
        h = np.ascontiguousarray(np.asarray(h, dtype=np.float64))
        n = (self.N + 1) * self.nq
        _chk(lib().idto_hip_constraint_schur_begin(self.h, dofs.ctypes.data_as(C.POINTER(C.c_int)), int(dofs.size)))
        lam, step, jtl = np.empty(h.size), np.empty(n), np.empty(n)
        rc = lib().idto_hip_constraint_solve(self.h, dptr(h), dptr(lam), dptr(step), dptr(jtl))
        if rc not in (0, 1):
            _chk(rc)
        return rc == 0, lam, step, jtl

    def constraint_step(self, lam):
        """H^-1 (g + J^T lambda) and J^T lambda, from the factors kept by constraint_schur"""
        lam = np.ascontiguousarray(np.asarray(lam, dtype=np.float64))
        n = (self.N + 1) * self.nq
        step, jtl = np.empty(n), np.empty(n)
        _chk(lib().idto_hip_constraint_step(self.h, dptr(lam), dptr(step), dptr(jtl)))
        return step, jtl

    def trial_cost(self, q):
        """upload q, evaluate tau and the cost, return (tau (N, nv), cost) with one synchronisation"""
        q = np.ascontiguousarray(np.asarray(q, dtype=np.float64))
        assert q.size == (self.N + 1) * self.nq
        tau = np.empty((self.N, self.nv))
        cost = C.c_double()
        _chk(lib().idto_hip_trial_cost(self.h, dptr(q), dptr(tau), C.byref(cost)))
        return tau, cost.value

    def eval_partials(self):
        _chk(lib().idto_hip_eval_partials(self.h))

    def grad_hess(self):
        _chk(lib().idto_hip_grad_hess(self.h))

    def factor_solve(self, rhs_ptr: int | None = None, nrhs: int = 1, x_ptr: int | None = None):
        _chk(lib().idto_hip_factor_solve(self.h, C.c_void_p(rhs_ptr) if rhs_ptr else None, int(nrhs),
                                         C.c_void_p(x_ptr) if x_ptr else None))

    def solve_host(self, rhs):
        """H X = rhs for host right-hand sides [nrhs, (N+1)*nq]"""
        rhs = np.ascontiguousarray(np.atleast_2d(np.asarray(rhs, dtype=np.float64)))
        x = np.zeros_like(rhs)
        _chk(lib().idto_hip_solve_host(self.h, dptr(rhs), rhs.shape[0], dptr(x)))
        return x

    def solve_dense_ldlt(self, rhs):
        """H x = rhs by a dense LDL^T of MakeDense() (SolverParameters::linear_solver = kDenseLdlt, TO.cc:2088-2093)"""
        rhs = np.ascontiguousarray(np.asarray(rhs, dtype=np.float64).ravel())
        x = np.zeros_like(rhs)
        _chk(lib().idto_hip_solve_dense_ldlt(self.h, dptr(rhs), dptr(x)))
        return x

    def set_option(self, name: str, value: int):
        _chk(lib().idto_hip_set_option(self.h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int()
        _chk(lib().idto_hip_get_option(self.h, name.encode(), C.byref(v)))
        return v.value

    def gn_step(self):
        _chk(lib().idto_hip_gn_step(self.h))

    def sync(self):
        _chk(lib().idto_hip_sync(self.h))

    def solver_status(self):
        """(failed, failed_rows_total) of the most recent factorisation; synchronises"""
        f, n = C.c_int(), C.c_int()
        _chk(lib().idto_hip_solver_status(self.h, C.byref(f), C.byref(n)))
        return bool(f.value), n.value

    # ---- timing (HIP events on the context's stream)
    def timing_enable(self, on=True):
        """True / 1: time every launch; s > 1: every s-th launch; False / 0: off"""
        _chk(lib().idto_hip_timing_enable(self.h, int(on)))

    def timing_reset(self):
        _chk(lib().idto_hip_timing_reset(self.h))

    def timing_get(self, which: int):
        ms, n = C.c_double(), C.c_int()
        _chk(lib().idto_hip_timing_get(self.h, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # ---- outputs
    def array_size(self, name):
        return lib().idto_hip_array_size(self.h, ARR[name])

    def device_ptr(self, name):
        return lib().idto_hip_device_ptr(self.h, ARR[name])

    @property
    def slab_stride(self):
        return lib().idto_hip_slab_stride(self.h)

    def prefetch(self, name):
        """enqueue the copy of a contiguous array now; the next get(name) waits only for it"""
        _chk(lib().idto_hip_prefetch(self.h, ARR[name]))

    def get(self, name, problem=None):
        """array `name` of problem 0 (through the prefetch staging if one is pending), or of problem
        `problem` of a batch"""
        out = np.zeros(self.array_size(name))
        if problem is None:
            _chk(lib().idto_hip_get(self.h, ARR[name], dptr(out)))
        else:
            _chk(lib().idto_hip_get_batch(self.h, ARR[name], int(problem), dptr(out)))
        N, nq, nv = self.N, self.nq, self.nv
        if name == "q":
            return out.reshape(N + 1, nq)
        if name == "v":
            return out.reshape(N + 1, nv)
        if name in ("a", "tau"):
            return out.reshape(N, nv)
        if name == "nplus":
            return out.reshape(N + 1, nq, nv).transpose(0, 2, 1).copy()
        if name.startswith("dtau"):
            return out.reshape(N, nq, nv).transpose(0, 2, 1).copy()  # [t, row(nv), col(nq)]
        if name.startswith("H_"):
            return out.reshape(N + 1, nq, nq).transpose(0, 2, 1).copy()  # [blk, row, col]
        if name == "cost":
            return float(out[0])
        return out


def rccl_info():
    """(path of the librccl this process resolved, its version code)"""
    buf, ver = C.create_string_buffer(1024), C.c_int(0)
    lib().idto_hip_rccl_info.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    _chk(lib().idto_hip_rccl_info(buf, 1024, C.byref(ver)))
    return buf.value.decode(), int(ver.value)


def comm_unique_id() -> bytes:
    """rank 0: the 128-byte id every rank passes to HipPath.comm_init"""
    buf = C.create_string_buffer(128)
    _chk(lib().idto_hip_comm_unique_id(buf, 128))
    return buf.raw


def comm_init_all(paths):
    """one process, one HipPath per device: ncclCommInitAll"""
    arr = (C.c_void_p * len(paths))(*[p.h for p in paths])
    _chk(lib().idto_hip_comm_init_all(arr, len(paths)))


def gn_step_multi(paths):
    arr = (C.c_void_p * len(paths))(*[p.h for p in paths])
    _chk(lib().idto_hip_gn_step_multi(arr, len(paths)))


def math_probe(x, device=0):
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    outs = [np.zeros_like(x) for _ in range(6)]
    _chk(lib().idto_hip_math_probe(device, dptr(x), x.size, *[dptr(o) for o in outs]))
    return dict(zip(("sqrt", "recip", "sin", "cos", "exp", "log"), outs))
