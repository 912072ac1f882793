"""Model-predictive-control shell around the optimizer: the caller of the hot path in the
reference's closed-loop examples (reference examples/mpc_controller.{h,cc}): at every replan
the last solution, stored as cubic splines, is time-shifted into the initial guess, the nominal
trajectory is shifted for the DoFs marked `q_nom_relative_to_q_init`, the initial condition is
reset and `SolveFromWarmStart` runs `mpc_iters` iterations from the previous trust-region
radius.  Drake's LeafSystem plumbing (ports, abstract state) is replaced by plain method calls.

Two implementations of the same shell:
  * `DeviceModelPredictiveController` - the product: the C++ class of include/idto/examples/mpc_controller.h inside
    libidto_opt.so through the C-ABI (idto_mpc_* of include/idto_opt.h);
  * `ModelPredictiveController` / `Interpolator` - the same logic in numpy / scipy over the Python optimizer binding; the
    tests hold the C++ shell against it (scipy's not-a-knot CubicSpline is the independent spline).
Both clamp the query time to the stored trajectory's time range, as Drake's PiecewisePolynomial::value does.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
from scipy.interpolate import CubicSpline


@dataclass
class StoredTrajectory:
    """reference examples/mpc_controller.h:43-55"""
    start_time: float = -1.0
    q: CubicSpline = None   # generalized positions
    v: CubicSpline = None   # generalized velocities
    u: CubicSpline = None   # control torques (actuated DoFs)


class ModelPredictiveController:
    """reference examples/mpc_controller.cc:13-138.  `optimizer` is an
    idto_amd.optimizer.TrajectoryOptimizer (or anything with the same methods)."""

    def __init__(self, optimizer, warm_start_solution, actuated=None):
        self.opt = optimizer
        self.time_step = optimizer.time_step()
        self.num_steps = optimizer.num_steps() + 1          # knots, as in the reference (:16)
        prob = optimizer.prob()
        self.nq, self.nv = len(prob.q_init), len(prob.v_init)
        act = np.ones(self.nv, bool) if actuated is None else np.asarray(actuated, bool)
        self.actuated = act if act.any() else np.ones(self.nv, bool)   # B = I without actuators
        self.warm_start = optimizer.CreateWarmStart(np.asarray(warm_start_solution.q))
        self.stored = StoredTrajectory()
        self.store_optimizer_solution(warm_start_solution, 0.0, self.stored)
        self.last_stats = None

    # ModelPredictiveController::UpdateAbstractState (:43-85)
    def update(self, t: float, q0, v0) -> StoredTrajectory:
        from .optimizer import TrajectoryOptimizerSolution, TrajectoryOptimizerStats
        q0, v0 = np.asarray(q0, float), np.asarray(v0, float)
        params, prob = self.opt.params(), self.opt.prob()
        sel = np.asarray(params.q_nom_relative_to_q_init, float)
        if sel.size != self.nq:
            raise ValueError("q_nom_relative_to_q_init must have one entry per position (mpc_controller.cc:45)")
        q_guess = self.update_initial_guess(self.stored, t)
        q_guess[0] = q0                      # the guess must be consistent with the initial condition
        self._last_guess = np.array(q_guess)
        self.warm_start.set_q(q_guess)
        q_nom = np.asarray(prob.q_nom, float)
        q_nom_new = q_nom + sel * (q0 - q_nom[0])           # (:62-69)
        self.opt.UpdateNominalTrajectory(q_nom_new, np.asarray(prob.v_nom, float))
        self.opt.ResetInitialConditions(q0, v0)
        sol, stats = TrajectoryOptimizerSolution(), TrajectoryOptimizerStats()
        self.opt.SolveFromWarmStart(self.warm_start, sol, stats)
        self.last_stats = stats
        self.store_optimizer_solution(sol, t, self.stored)
        return self.stored

    # UpdateInitialGuess (:87-97)
    def update_initial_guess(self, stored: StoredTrajectory, current_time: float):
        start = current_time - stored.start_time
        ts = start + self.time_step * np.arange(self.num_steps)
        return np.asarray(stored.q(_clamp(stored.q, ts)), float)

    # StoreOptimizerSolution (:99-138): cubic splines with continuous second derivatives
    # (Drake's default end condition is not-a-knot, scipy's default too)
    def store_optimizer_solution(self, solution, start_time: float, stored: StoredTrajectory):
        q, v, tau = (np.asarray(x, float) for x in (solution.q, solution.v, solution.tau))
        n = self.num_steps
        ts = self.time_step * np.arange(n)
        u = np.vstack([tau, tau[-1:]])[:, self.actuated]    # undefined at the last step: repeat (:122-126)
        stored.start_time = float(start_time)
        bc = "not-a-knot" if n >= 4 else "natural"
        stored.q = CubicSpline(ts, q[:n], bc_type=bc)
        stored.v = CubicSpline(ts, v[:n], bc_type=bc)
        stored.u = CubicSpline(ts, u[:n], bc_type=bc)


def _clamp(spline, t):
    """PiecewisePolynomial::value evaluates at the closest point of the trajectory's time range"""
    return np.clip(t, spline.x[0], spline.x[-1])


class Interpolator:
    """reference examples/mpc_controller.cc:140-178: x(t) and u(t) of a stored trajectory"""

    @staticmethod
    def state(traj: StoredTrajectory, t: float):
        s = t - traj.start_time
        return np.concatenate([traj.q(_clamp(traj.q, s)), traj.v(_clamp(traj.v, s))])

    @staticmethod
    def control(traj: StoredTrajectory, t: float):
        return np.asarray(traj.u(_clamp(traj.u, t - traj.start_time)), float)


class DeviceModelPredictiveController:
    """idto::examples::mpc::ModelPredictiveController + Interpolator of libidto_opt.so (include/idto_opt.h idto_mpc_*).
    `optimizer`: an idto_amd.optimizer.TrajectoryOptimizer whose max_iterations is the example's mpc_iters."""

    def __init__(self, optimizer, warm_start_solution, actuated=None, q_nom_relative_to_q_init=None, replan_period=0.0,
                 strict=True):
        """strict: update() raises when a re-plan's factorisation fails (the C++ controller keeps the previous plan and
        reports SolverFlag::kFactorizationFailed through last_flag(); a caller that ignores the flag would otherwise
        run on the old plan with no signal).  strict=False: return the previous plan, `last_flag == 2`."""
        import ctypes as C
        self.strict = bool(strict)
        from . import optimizer as O
        self._O, self._C = O, C
        self.opt = optimizer
        prob = optimizer.prob()
        self.N, self.nq, self.nv = optimizer.num_steps(), len(prob.q_init), len(prob.v_init)
        q, v, tau = (O._d(x) for x in (warm_start_solution.q, warm_start_solution.v, warm_start_solution.tau))
        act = None if actuated is None else np.ascontiguousarray(np.asarray(actuated, dtype=np.int32))
        sel = q_nom_relative_to_q_init
        if sel is None:
            sel = optimizer.params().q_nom_relative_to_q_init
        sel = None if sel is None or len(sel) == 0 else np.ascontiguousarray(np.asarray(sel, dtype=np.int32))
        ip = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_int))
        h = C.c_void_p()
        self._chk(O.lib().idto_mpc_create(optimizer._h, O.dptr(q), O.dptr(v), O.dptr(tau), ip(act), ip(sel),
                                          float(replan_period), C.byref(h)))
        self._h = h
        self.nu = O.lib().idto_mpc_num_actuators(self._h)
        self.last_cost = None

    def _chk(self, rc):
        if rc:
            raise RuntimeError(self._O.lib().idto_opt_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            self._O.lib().idto_mpc_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _buffers(self):
        """the re-plan's input / output buffers and their ctypes pointers, made once: a re-plan is ~0.2 ms, five fresh
        arrays, five pointer casts and two byref objects per call were 15 - 20 us of it on the Python side"""
        O, C = self._O, self._C
        b = getattr(self, "_buf", None)
        if b is None:
            x0 = np.zeros(self.nq + self.nv)
            g, q = np.zeros((self.N + 1, self.nq)), np.zeros((self.N + 1, self.nq))
            v, tau = np.zeros((self.N + 1, self.nv)), np.zeros((self.N, self.nv))
            cost, flag = C.c_double(), C.c_int()
            ptrs = tuple(O.dptr(a) for a in (x0, g, q, v, tau)) + (C.byref(cost), C.byref(flag))
            b = self._buf = (x0, g, q, v, tau, cost, flag, ptrs, O.lib().idto_mpc_update)
        return b

    def update(self, t, q0, v0, copy=True):
        """UpdateAbstractState: returns (q_guess, q, v, tau) of this replan.  copy=False: the controller's own output
        buffers, overwritten by the next update (what a C++ caller of idto_mpc_update holds)"""
        x0, g, q, v, tau, cost, flag, ptrs, fn = self._buffers()
        x0[:self.nq] = q0
        x0[self.nq:] = v0
        if fn(self._h, t, *ptrs):
            raise RuntimeError(self._O.lib().idto_opt_last_error().decode())
        self.last_cost = cost.value
        self.last_flag = flag.value
        if flag.value == 2 and self.strict:   # SolverFlag::kFactorizationFailed: (q, v, tau) are the PREVIOUS plan's
            raise RuntimeError("MPC re-plan at t = %g: the factorisation failed; the previous plan stays in force "
                               "(strict=False returns it, last_flag == 2 says so)" % t)
        return (g.copy(), q.copy(), v.copy(), tau.copy()) if copy else (g, q, v, tau)

    @property
    def start_time(self):
        return self._O.lib().idto_mpc_start_time(self._h)

    def state(self, t):
        x = np.zeros(self.nq + self.nv)
        self._chk(self._O.lib().idto_mpc_state(self._h, float(t), self._O.dptr(x)))
        return x

    def control(self, t):
        u = np.zeros(self.nu)
        self._chk(self._O.lib().idto_mpc_control(self._h, float(t), self._O.dptr(u)))
        return u


def spline_eval(breaks, knots, times):
    """PiecewiseCubic of include/idto/examples/mpc_controller.h (C++, host only): values at `times`"""
    from . import optimizer as O
    breaks, knots, times = O._d(breaks), O._d(knots), O._d(np.atleast_1d(times))
    n, dim = knots.shape
    out = np.zeros((len(times), dim))
    if O.lib().idto_mpc_spline_eval(O.dptr(breaks), O.dptr(knots), n, dim, O.dptr(times), len(times), O.dptr(out)):
        raise RuntimeError(O.lib().idto_opt_last_error().decode())
    return out
