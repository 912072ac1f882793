#!/usr/bin/env python3
"""Host-side timeline of ONE MPC re-plan: what the host thread does, mark by mark, inside
`idto_mpc_update` = reference examples/mpc_controller.cc:43-85 UpdateAbstractState (shift the stored solution,
SolveFromWarmStart with the example's mpc_iters, store the splines), with the C++ controller of
include/idto/examples/mpc_controller.h on the device.  The marks are idto_hip_trace_mark's (include/idto_hip.h), set in
libidto_hip.so and libidto_opt.so; the timeline is the median over `reps` re-plans of every interval.
Usage: python tools/mpc_timeline.py [model ...]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from idto_amd import hip  # noqa: E402
from idto_amd.model import load_model  # noqa: E402
from idto_amd.mpc import DeviceModelPredictiveController  # noqa: E402
from idto_amd.optimizer import TrajectoryOptimizer, TrajectoryOptimizerSolution, TrajectoryOptimizerStats  # noqa: E402
from idto_amd.problem import SolverParameters, load_config, make_problem  # noqa: E402

L = hip.lib()
L.idto_hip_trace_enable.argtypes = [C.c_int]
L.idto_hip_trace_enable.restype = None
L.idto_hip_trace_dump.argtypes = [C.c_char_p, C.c_int]
L.idto_hip_trace_dump.restype = C.c_int


def dump():
    n = L.idto_hip_trace_dump(None, 0)
    buf = C.create_string_buffer(n + 8)
    L.idto_hip_trace_dump(buf, n + 8)
    out = []
    for line in buf.value.decode().splitlines():
        t, label = line.split(" ", 1)
        out.append((float(t), label))
    return out


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
    runs, walls = [], []
    for i in range(50):
        t = i * period
        x = mpc.state(t) if i else np.concatenate([q0, v0])
        L.idto_hip_trace_enable(1)
        t0 = time.perf_counter()
        mpc.update(t, x[:model.nq], x[model.nq:], copy=False)   # (the controller's own output buffers, as a C++ caller holds them)
        walls.append((time.perf_counter() - t0) * 1e6)
        ev = dump()
        L.idto_hip_trace_enable(0)
        if i >= 10:
            runs.append(ev)
    labels = [l for _, l in runs[0]]
    runs = [r for r in runs if [l for _, l in r] == labels]
    T = np.array([[t for t, _ in r] for r in runs])
    med = np.median(np.diff(np.concatenate([np.zeros((T.shape[0], 1)), T], axis=1), axis=1), axis=0)
    total = np.median(walls[10:])
    print(f"{name}: N={prob.num_steps}, mpc_iters {iters}, constraints {'enforced' if sp.equality_constraints else 'off'}: "
          f"Python's wall clock around mpc.update {total:.1f} us (median of {len(walls) - 10}); marks of {len(runs)} re-plans, median interval before each mark")
    acc = 0.0
    for lab, dt in zip(labels, med):
        acc += dt
        print(f"   {acc:8.1f} us  (+{dt:6.1f})  {lab}")
    print(f"   {total:8.1f} us  (+{total - acc:6.1f})  back in Python (ctypes return, numpy views of the outputs)", flush=True)
    mpc.close()
