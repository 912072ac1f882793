import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
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
    d = dev.get("debug")[:8] / 100.0
    print(name, "phases (us from the kernel's first stamp): loads %.2f, v/a/dq %.2f, evaluations %.2f, records %.2f, assembly %.2f, solve %.2f" % tuple(d[1:7] - d[0]))
