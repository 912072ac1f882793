"""Bit reproducibility of the resident trust-region loop (idto_hip_tr_solve) over repeated solves and fresh contexts:
tr_iter_kernel's workgroups hand their sums to each other with the launch's epoch in every word (trust_region.h), the small
models run the iteration in one launch (gn_small.h).  usage: stress_tr.py model N iterations solves [constrained]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory, SCALING
name, N, iters, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
con = len(sys.argv) > 5
cfg, model = load_config(name), load_model(name)
prob, sp, _ = make_problem(cfg, model, num_steps=N)
q = synthetic_trajectory(cfg, model, N, seed=0, lower=0.01)
dofs = model.unactuated_dofs if con else ()
want, bad, dev = None, 0, None
for s in range(count):
    if s % 50 == 0:
        if dev: dev.close()
        dev = hip.HipPath(model, prob, sp)
    dev.set_q(q)
    dev.eval_tau()
    rows, delta = dev.tr_solve(iters, SCALING["double_sqrt"], True, False, 1e-1, 1e5, constrained_dofs=dofs)
    got = (np.delete(rows, 10, axis=1), delta, dev.get("q"))
    if want is None: want = got
    elif not (np.array_equal(got[0], want[0]) and got[1] == want[1] and np.array_equal(got[2], want[2])): bad += 1
print(f"{name} N={N} {'constrained ' if con else ''}: {count} solves of {iters} iterations over {(count + 49) // 50} contexts, "
      f"{int(want[0][:, 9].sum())} accepted steps each, {bad} solves with different bits, timeouts {dev.get_option('solver_timeouts')}")
