"""Model-predictive-control shell around the optimizer: the caller of the hot path in the
reference's closed-loop examples (reference examples/mpc_controller.{h,cc}): at every replan
the last solution, stored as cubic splines, is time-shifted into the initial guess, the nominal
trajectory is shifted for the DoFs marked `q_nom_relative_to_q_init`, the initial condition is
reset and `SolveFromWarmStart` runs `mpc_iters` iterations from the previous trust-region
radius.  Drake's LeafSystem plumbing (ports, abstract state) is replaced by plain method calls.
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
        return np.asarray(stored.q(ts), float)

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


class Interpolator:
    """reference examples/mpc_controller.cc:140-178: x(t) and u(t) of a stored trajectory"""

    @staticmethod
    def state(traj: StoredTrajectory, t: float):
        s = t - traj.start_time
        return np.concatenate([traj.q(s), traj.v(s)])

    @staticmethod
    def control(traj: StoredTrajectory, t: float):
        return np.asarray(traj.u(t - traj.start_time), float)
