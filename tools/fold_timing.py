"""Kernel times of the finite-difference and assembly launches with the assembly fold on and off
(HIP events on the context's stream, include/idto_hip.h idto_hip_timing_*)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory

name, N = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("mini_cheetah", 40)
cfg, model = load_config(name), load_model(name)
prob, sp, _ = make_problem(cfg, model, num_steps=N)
sp.scaling = False
sp.equality_constraints = False
q = synthetic_trajectory(cfg, model, N, seed=0, lower=0.01)
for fold, stop in ((0, 0), (1, 0), (1, 1), (1, 2), (1, 3), (1, 4), (1, 5), (1, 6), (1, 7)):
    dev = hip.HipPath(model, prob, sp)
    dev.set_option("asm_fold", fold)
    dev.set_option("fd_stop", stop)
    dev.set_q(q)
    for _ in range(20):
        dev.gn_step()
    dev.sync()
    dev.timing_enable(True)
    dev.timing_reset()
    for _ in range(100):
        dev.gn_step()
    dev.sync()
    t = [dev.timing_get(i) for i in range(3)]
    print(f"fold={fold} stop={stop} last_assembly={dev.get_option('last_assembly')} fd={t[0][0]*1e3:.2f}us asm={t[1][0]*1e3:.2f}us "
          f"solve={t[2][0]*1e3:.2f}us", flush=True)
    dev.close()

