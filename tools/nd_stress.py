"""Stress for the flag / counter protocol of the nested-dissection solver: the trajectory changes
every launch, so any read that is not ordered after its producer returns the PREVIOUS launch's
(different) values and shows up as a wrong step."""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory
name, N, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg, model = load_config(name), load_model(name)
prob, sp, _ = make_problem(cfg, model, num_steps=N)
sp.scaling = False; sp.equality_constraints = False
qs = [synthetic_trajectory(cfg, model, N, seed=s, lower=0.01) for s in range(4)]
dev = hip.HipPath(model, prob, sp)
dev.set_option("solver_nd", 0)
ref = []
for q in qs:
    dev.set_q(q); dev.gn_step(); ref.append(dev.get("step").copy())
dev.set_option("solver_nd", int(os.environ.get("ND", "1")))
bad = 0
for it in range(iters):
    j = it % 4
    dev.set_q(qs[j]); dev.gn_step()
    try:
        p = dev.get("step")
    except hip.FactorizationFailed:
        p = np.full_like(ref[j], np.nan)
    e = np.abs(p - ref[j]).reshape(N + 1, model.nq).max(axis=1) / np.abs(ref[j]).max()
    if not (e.max() < 1e-4):
        bad += 1
        if bad <= 8:
            print(it, "wrong: max rel diff %.2e" % np.nanmax(e), "rows:", np.where(~(e < 1e-4))[0][:10], "nan" if np.isnan(e).any() else "")
print(name, N, "wrong launches:", bad, "of", iters)
