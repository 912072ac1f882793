"""`TrajectoryOptimizer` — Python face of the host-side optimizer (libidto_opt.so,
include/idto_opt.h), the counterpart of the reference's pybind11 class
(reference python_bindings/trajectory_optimizer_py.cc:34-59): same method names
(`time_step`, `num_steps`, `Solve`, `CreateWarmStart`, `SolveFromWarmStart`,
`ResetInitialConditions`, `UpdateNominalTrajectory`, `params`, `prob`), with the Drake
plant replaced by a `Model`.  The hot path runs on the MI355X (libidto_hip.so); there is
no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import hip
from .model import CContactParams, CModel, CProblem, CSolverParams, CStats, Model, dptr, iptr
from .problem import ProblemDefinition, SolverParameters

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libidto_opt.so")
_lib = None

SOLVER_FLAGS = ("kSuccess", "kLinesearchMaxIters", "kFactorizationFailed", "kMaxIterationsReached")


def lib():
    global _lib
    if _lib is None:
        hip.lib()  # loads torch's HIP runtime first, then libidto_hip.so (see idto_amd/hip.py)
        if not os.path.exists(LIB_PATH):
            raise hip.HipLibraryMissing(f"{LIB_PATH} not found: run ./build.sh")
        L = C.CDLL(LIB_PATH)
        L.idto_opt_last_error.restype = C.c_char_p
        L.idto_opt_dense_ldlt_solve.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double)]
        L.idto_opt_create.argtypes = [C.POINTER(CModel), C.POINTER(CProblem), C.POINTER(CContactParams),
                                      C.POINTER(CSolverParams), C.c_int, C.POINTER(C.c_void_p)]
        L.idto_opt_create_multi.argtypes = [C.POINTER(CModel), C.POINTER(CProblem), C.POINTER(CContactParams),
                                            C.POINTER(CSolverParams), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
        L.idto_opt_destroy.argtypes = [C.c_void_p]
        L.idto_opt_num_steps.argtypes = [C.c_void_p]
        L.idto_opt_time_step.argtypes = [C.c_void_p]
        L.idto_opt_time_step.restype = C.c_double
        L.idto_opt_num_equality_constraints.argtypes = [C.c_void_p]
        P = C.POINTER(C.c_double)
        L.idto_opt_solve.argtypes = [C.c_void_p, P, P, P, P, C.POINTER(CStats), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.idto_opt_ws_create.argtypes = [C.c_void_p, P, C.POINTER(C.c_void_p)]
        L.idto_opt_ws_destroy.argtypes = [C.c_void_p]
        L.idto_opt_ws_set_q.argtypes = [C.c_void_p, C.c_void_p, P]
        L.idto_opt_ws_get.argtypes = [C.c_void_p, C.c_void_p, P, P]
        L.idto_opt_ws_solve.argtypes = [C.c_void_p, C.c_void_p, P, P, P, C.POINTER(CStats), C.POINTER(C.c_int),
                                        C.POINTER(C.c_int)]
        L.idto_opt_reset_initial_conditions.argtypes = [C.c_void_p, P, P]
        L.idto_opt_update_nominal_trajectory.argtypes = [C.c_void_p, P, P]
        L.idto_opt_eval.argtypes = [C.c_void_p, P, P, P, P, P, P, P, P]
        L.idto_opt_dogleg.argtypes = [C.c_void_p, P, C.c_double, P, P, C.POINTER(C.c_int)]
        L.idto_opt_trust_ratio.argtypes = [C.c_void_p, P, P, P]
        I = C.POINTER(C.c_int)
        L.idto_mpc_create.argtypes = [C.c_void_p, P, P, P, I, I, C.c_double, C.POINTER(C.c_void_p)]
        L.idto_mpc_destroy.argtypes = [C.c_void_p]
        L.idto_mpc_num_actuators.argtypes = [C.c_void_p]
        L.idto_mpc_update.argtypes = [C.c_void_p, C.c_double, P, P, P, P, P, P, I]
        L.idto_mpc_state.argtypes = [C.c_void_p, C.c_double, P]
        L.idto_mpc_control.argtypes = [C.c_void_p, C.c_double, P]
        L.idto_mpc_start_time.argtypes = [C.c_void_p]
        L.idto_mpc_start_time.restype = C.c_double
        L.idto_mpc_spline_eval.argtypes = [P, P, C.c_int, C.c_int, P, C.c_int, P]
        _lib = L
    return _lib


EXPORTED_SYMBOLS = [
    "idto_opt_last_error", "idto_opt_dense_ldlt_solve", "idto_opt_create", "idto_opt_create_multi", "idto_opt_destroy", "idto_opt_num_steps", "idto_opt_time_step",
    "idto_opt_num_equality_constraints", "idto_opt_solve", "idto_opt_ws_create", "idto_opt_ws_destroy",
    "idto_opt_ws_set_q", "idto_opt_ws_get", "idto_opt_ws_solve", "idto_opt_reset_initial_conditions",
    "idto_opt_update_nominal_trajectory", "idto_opt_eval", "idto_opt_dogleg", "idto_opt_trust_ratio",
]
MPC_SYMBOLS = ["idto_mpc_create", "idto_mpc_destroy", "idto_mpc_num_actuators", "idto_mpc_update", "idto_mpc_state",
               "idto_mpc_control", "idto_mpc_start_time", "idto_mpc_spline_eval"]


def _d(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


class TrajectoryOptimizerStats:
    """reference optimizer/trajectory_optimizer_solution.h:58-185"""
    FIELDS = ["iteration_times", "iteration_costs", "linesearch_iterations", "linesearch_alphas",
              "trust_region_radii", "q_norms", "dq_norms", "dqH_norms", "trust_ratios", "gradient_norms", "dL_dqs",
              "h_norms", "merits"]

    def __init__(self, capacity=1000):
        self.c = CStats()
        self._arr = {}
        self._reserve(capacity)

    def _reserve(self, capacity):
        """room for `capacity` iterations (Solve sizes it from SolverParameters::max_iterations)"""
        assert self.c.count == 0
        self.c.capacity = capacity
        for f in self.FIELDS:
            a = np.zeros(capacity, dtype=np.int32 if f == "linesearch_iterations" else np.float64)
            setattr(self.c, f, iptr(a) if f == "linesearch_iterations" else dptr(a))
            self._arr[f] = a

    @property
    def truncated(self):
        """True if the solver ran more iterations than the arrays could hold"""
        return self.c.total > self.c.count

    def __getattr__(self, k):
        if k in TrajectoryOptimizerStats.FIELDS:
            return self._arr[k][: self.c.count]
        raise AttributeError(k)

    @property
    def solve_time(self):
        return self.c.solve_time

    def is_empty(self):
        return self.c.count == 0


class TrajectoryOptimizerSolution:
    """reference optimizer/trajectory_optimizer_solution.h:43-52"""

    def __init__(self):
        self.q = self.v = self.tau = None


class WarmStart:
    def __init__(self, opt, handle):
        self._opt, self._h = opt, handle

    def set_q(self, q):
        opt = self._opt
        opt._chk(lib().idto_opt_ws_set_q(opt._h, self._h, dptr(_d(q))))

    def get_q(self):
        opt = self._opt
        q = np.zeros((opt.N + 1, opt.nq))
        opt._chk(lib().idto_opt_ws_get(opt._h, self._h, dptr(q), None))
        return q

    @property
    def Delta(self):
        d = C.c_double()
        self._opt._chk(lib().idto_opt_ws_get(self._opt._h, self._h, None, C.byref(d)))
        return d.value

    def __del__(self):
        try:
            lib().idto_opt_ws_destroy(self._h)
        except Exception:
            pass


class TrajectoryOptimizer:
    def __init__(self, model: Model, prob: ProblemDefinition, params: SolverParameters | None = None, device: int = 0,
                 devices=None):
        """`devices`: list of HIP devices of this node to shard the finite-difference grid over
        (devices[0] hosts the optimizer; one RCCL all-gather per evaluation of the partials)"""
        L = lib()
        params = params or SolverParameters()
        self.model, self._prob, self._params = model, prob, params
        self.nq, self.nv, self.N = model.nq, model.nv, prob.num_steps
        cm, self._k1 = model.to_c()
        cp, self._k2 = prob.to_c()
        cc, cs = params.contact_to_c(), params.to_c()
        h = C.c_void_p()
        if devices is None:
            rc = L.idto_opt_create(C.byref(cm), C.byref(cp), C.byref(cc), C.byref(cs), int(device), C.byref(h))
        else:  # sharded over several devices of the node (RCCL all-gather of the partials)
            arr = (C.c_int * len(devices))(*[int(d) for d in devices])
            rc = L.idto_opt_create_multi(C.byref(cm), C.byref(cp), C.byref(cc), C.byref(cs), arr, len(devices), C.byref(h))
        if rc != 0:
            raise RuntimeError(L.idto_opt_last_error().decode())
        self._h = h

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(lib().idto_opt_last_error().decode())

    def close(self):
        """release the device context now (otherwise when the object is collected)"""
        if getattr(self, "_h", None):
            lib().idto_opt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference API
    def time_step(self):
        return lib().idto_opt_time_step(self._h)

    def num_steps(self):
        return lib().idto_opt_num_steps(self._h)

    def num_equality_constraints(self):
        return lib().idto_opt_num_equality_constraints(self._h)

    def params(self):
        return self._params

    def prob(self):
        return self._prob

    def _outputs(self):
        return np.zeros((self.N + 1, self.nq)), np.zeros((self.N + 1, self.nv)), np.zeros((self.N, self.nv))

    def Solve(self, q_guess, solution: TrajectoryOptimizerSolution, stats: TrajectoryOptimizerStats):
        """returns the SolverFlag name; fills `solution` and `stats` (must be empty)"""
        if not stats.is_empty():
            raise RuntimeError("stats must be empty")
        if stats.c.capacity < self._params.max_iterations:
            stats._reserve(self._params.max_iterations)
        q, v, tau = self._outputs()
        flag, reason = C.c_int(), C.c_int()
        self._chk(lib().idto_opt_solve(self._h, dptr(_d(q_guess)), dptr(q), dptr(v), dptr(tau), C.byref(stats.c),
                                       C.byref(flag), C.byref(reason)))
        solution.q, solution.v, solution.tau = q, v, tau
        self.last_convergence_reason = reason.value
        return SOLVER_FLAGS[flag.value]

    def CreateWarmStart(self, q_guess):
        h = C.c_void_p()
        self._chk(lib().idto_opt_ws_create(self._h, dptr(_d(q_guess)), C.byref(h)))
        return WarmStart(self, h)

    def SolveFromWarmStart(self, warm_start: WarmStart, solution: TrajectoryOptimizerSolution,
                           stats: TrajectoryOptimizerStats):
        if stats.is_empty() and stats.c.capacity < self._params.max_iterations:
            stats._reserve(self._params.max_iterations)
        q, v, tau = self._outputs()
        flag, reason = C.c_int(), C.c_int()
        self._chk(lib().idto_opt_ws_solve(self._h, warm_start._h, dptr(q), dptr(v), dptr(tau), C.byref(stats.c),
                                          C.byref(flag), C.byref(reason)))
        solution.q, solution.v, solution.tau = q, v, tau
        self.last_convergence_reason = reason.value
        return SOLVER_FLAGS[flag.value]

    def ResetInitialConditions(self, q_init, v_init):
        self._chk(lib().idto_opt_reset_initial_conditions(self._h, dptr(_d(q_init)), dptr(_d(v_init))))
        self._prob.q_init, self._prob.v_init = np.array(q_init, float), np.array(v_init, float)

    def UpdateNominalTrajectory(self, q_nom, v_nom):
        self._chk(lib().idto_opt_update_nominal_trajectory(self._h, dptr(_d(q_nom)), dptr(_d(v_nom))))
        self._prob.q_nom, self._prob.v_nom = np.array(q_nom, float), np.array(v_nom, float)

    # ---- what the reference's C++ tests reach through TrajectoryOptimizerTester
    def eval(self, q):
        nvars, neq = (self.N + 1) * self.nq, self.num_equality_constraints()
        cost, merit = C.c_double(), C.c_double()
        out = dict(gradient=np.zeros(nvars), scaled_gradient=np.zeros(nvars), scale_factors=np.zeros(nvars),
                   merit_gradient=np.zeros(nvars))
        lam = np.zeros(max(neq, 1))
        use_eq = bool(self._params.equality_constraints) and neq > 0
        self._chk(lib().idto_opt_eval(self._h, dptr(_d(q)), C.byref(cost), dptr(out["gradient"]),
                                      dptr(out["scaled_gradient"]), dptr(out["scale_factors"]),
                                      dptr(lam) if use_eq else None, C.byref(merit), dptr(out["merit_gradient"])))
        out.update(cost=cost.value, merit=merit.value, lagrange_multipliers=lam[:neq] if use_eq else np.zeros(0))
        return out

    def dogleg(self, q, Delta):
        nvars = (self.N + 1) * self.nq
        dq, dqH, act = np.zeros(nvars), np.zeros(nvars), C.c_int()
        self._chk(lib().idto_opt_dogleg(self._h, dptr(_d(q)), float(Delta), dptr(dq), dptr(dqH), C.byref(act)))
        return dq, dqH, bool(act.value)

    def trust_ratio(self, q, dq):
        rho = C.c_double()
        self._chk(lib().idto_opt_trust_ratio(self._h, dptr(_d(q)), dptr(_d(dq)), C.byref(rho)))
        return rho.value
