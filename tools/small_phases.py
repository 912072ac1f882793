"""Phases of gn_small_kernel (csrc/gn_small.h) by its own wall-clock stamps: IDTO_SMALL_STAMPS=1 python tools/small_phases.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory
for name in ("acrobot", "spinner"):
    cfg, model = load_config(name), load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=40)
    sp.scaling = sp.equality_constraints = False
    dev = hip.HipPath(model, prob, sp)
    dev.set_q(synthetic_trajectory(cfg, model, 40, seed=0, lower=0.0))
    for _ in range(5): dev.gn_step()
    dev.sync()
    raw = dev.get("debug")
    d = raw[:8] / 100.0
    b = raw[8:13] / 100.0 - d[0]
    print(name, "  inside the solve: begins %.2f, (padding) %.2f, copies staged %.2f, forward done %.2f, back substitution done %.2f" % tuple(b))
    print(name, "phases (us from the kernel's first stamp): loads %.2f, v/a/dq %.2f, evaluations %.2f, records %.2f, assembly %.2f, solve %.2f" % tuple(d[1:7] - d[0]))
