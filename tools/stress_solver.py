#!/usr/bin/env python3
"""Repeats the Gauss-Newton step many times on fresh contexts and fixed inputs: every launch must reproduce the first
launch's bits and report a clean factorisation status (a race between the solver's workgroups shows up here)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory

name = sys.argv[1] if len(sys.argv) > 1 else "mini_cheetah"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
cfg, model = load_config(name), load_model(name)
prob, sp, _ = make_problem(cfg, model, num_steps=N)
sp.scaling = sp.equality_constraints = False
q = synthetic_trajectory(cfg, model, N, seed=0, lower=0.01)
for inside in (1, 0):
    bad = mism = 0
    ref = None
    for ctx in range(6):
        dev = hip.HipPath(model, prob, sp)
        dev.set_option("asm_in_solver", inside)
        dev.set_q(q)
        for i in range(reps // 6):
            dev.gn_step()
            try:
                x = dev.get("step")
            except hip.FactorizationFailed:
                bad += 1
                continue
            if ref is None:
                ref = x
            elif not np.array_equal(x, ref):
                mism += 1
        dev.close()
    print(f"{name} N={N} asm_in_solver={inside}: {reps} launches, {bad} flagged factorisations, {mism} results that differ from the first")
