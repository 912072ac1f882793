"""The scalar band factorisation (csrc/penta_band.h) on the small models: step time with it and with the pipelined block
kernel, and the phases of its launch from the kernel's own stamps (option "solver_debug"; 100 MHz wall clock)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory

for name, N in (("acrobot", 40), ("spinner", 40), ("hopper", 50)):
    cfg, model = load_config(name), load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    sp.scaling = False; sp.equality_constraints = False
    q = synthetic_trajectory(cfg, model, N, seed=0, lower=0.01)
    line = []
    for label, band in (("band", 2), ("pipelined block kernel", 0)):
        dev = hip.HipPath(model, prob, sp)
        dev.set_option("solver_band", band)
        dev.set_q(q)
        for _ in range(20): dev.gn_step()
        solver = dev.get_option("last_solver")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(500): dev.gn_step()
        dev.get("step")
        line.append(f"{label} (last_solver {solver}) {(time.perf_counter() - t0) / 500 * 1e6:.1f} us/step")
        if band:
            dev.set_option("solver_debug", 1)
            dev.gn_step(); dev.gn_step()
            d = dev.get("debug")[:8] / 100.0
            ph = "launch start -> g, H assembled (workgroups of the same launch) %.2f -> copies of the band in LDS %.2f -> forward done (both chains, join, middle rows) %.2f -> back substitution done %.2f us" % tuple(d[1:5] - d[0])
        dev.close()
    print(f"{name} N={N} (block {model.nq}, half width {3 * model.nq - 1}, {N * model.nq} pivots): " + "; ".join(line))
    print("   " + ph)
