"""The device-resident trust-region loop with the example's equality constraints (idto_hip_tr_solve): time per
iteration, flags / costs / decisions of every iteration; CON_STAMPS=1 adds the cycle stamps of the single-workgroup
multiplier solve (constraints.h constraint_lambda_kernel; option solver_debug).
  python tools/constrained_loop.py hopper 40 20"""
import sys, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, SCALING
name, N, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg, model = load_config(name), load_model(name)
prob, sp, q_guess = make_problem(cfg, model, num_steps=N)
dev = hip.HipPath(model, prob, sp)
dev.set_unactuated_dofs(model.unactuated_dofs)
for rep in range(2):
    dev.set_q(np.asarray(q_guess))
    dev.eval_tau()
    t0 = time.perf_counter()
    rows, delta = dev.tr_solve(iters, SCALING[sp.scaling_method] if sp.scaling else -1, sp.scaling, False, sp.Delta0, sp.Delta_max,
                               constrained_dofs=model.unactuated_dofs)
    dt = time.perf_counter() - t0
print(name, "neq", len(model.unactuated_dofs) * N, f"{1e3*dt/iters:.4f} ms/iter")
print("flags", rows[:, 14].astype(int))
print("cost", rows[:6, 0], "rho", rows[:6, 2], "acc", rows[:, 9].astype(int))
print("clock diffs us", np.diff(rows[:, 10])[:10] * 0.01)
if os.environ.get("CON_STAMPS"):
    dev.set_option("solver_debug", 1)
    dev.set_option("solver_nd", 0)
    dev.set_q(np.asarray(q_guess)); dev.eval_tau()
    dev.tr_solve(2, SCALING[sp.scaling_method] if sp.scaling else -1, sp.scaling, False, sp.Delta0, sp.Delta_max,
                 constrained_dofs=model.unactuated_dofs)
    st = dev.get("debug")
    print("multiplier solve, cycles: factorisation + forward substitution", st[2] - st[0], " backward substitution", st[3] - st[2])
