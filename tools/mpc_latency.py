#!/usr/bin/env python3
"""Wall-clock latency of one MPC re-plan (reference examples/mpc_controller.cc:43-85 UpdateAbstractState: shift the stored
solution, SolveFromWarmStart with the example's mpc_iters, store the splines) with the C++ controller of
include/idto/examples/mpc_controller.h on the device.  Usage: python tools/mpc_latency.py [model ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from idto_amd.model import load_model  # noqa: E402
from idto_amd.mpc import DeviceModelPredictiveController  # noqa: E402
from idto_amd.optimizer import TrajectoryOptimizer, TrajectoryOptimizerSolution, TrajectoryOptimizerStats  # noqa: E402
from idto_amd.problem import SolverParameters, load_config, make_problem  # noqa: E402

for name in (sys.argv[1:] or ["mini_cheetah", "hopper", "spinner", "allegro_hand"]):
    cfg, model = load_config(name), load_model(name)
    prob, sp, q_guess = make_problem(cfg, model)
    sp.verbose = False
    sp.max_iterations = min(int(sp.max_iterations), 30)
    opt = TrajectoryOptimizer(model, prob, sp)
    sol, st = TrajectoryOptimizerSolution(), TrajectoryOptimizerStats()
    opt.Solve(q_guess, sol, st)
    iters = int(cfg.get("mpc_iters", 1))
    sp1 = SolverParameters(**{**sp.__dict__, "max_iterations": iters})
    period = 1.0 / float(cfg.get("controller_frequency", 200.0))
    opt1 = TrajectoryOptimizer(model, prob, sp1)
    mpc = DeviceModelPredictiveController(opt1, sol, actuated=model.actuated, replan_period=period)
    q0, v0 = np.asarray(sol.q[0]).copy(), np.asarray(sol.v[0]).copy()
    times = []
    for i in range(60):
        t = i * period
        x = mpc.state(t) if i else np.concatenate([q0, v0])
        t0 = time.perf_counter()
        mpc.update(t, x[:model.nq], x[model.nq:], copy=False)   # (the controller's own output buffers, as a C++ caller holds them)
        times.append(time.perf_counter() - t0)
    ts = np.sort(np.array(times[10:])) * 1e3
    print(f"{name}: N={prob.num_steps}, mpc_iters {iters}, constraints {'enforced' if sp.equality_constraints else 'off'}: re-plan "
          f"median {np.median(ts):.3f} ms, p10 {ts[len(ts) // 10]:.3f}, p90 {ts[9 * len(ts) // 10]:.3f} (controller period {1e3 * period:.2f} ms)", flush=True)
    mpc.close()
