import sys, os, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory
import oracle_lib as ol
from oracle_lib import Oracle
name, N = sys.argv[1], int(sys.argv[2])
cfg, model = load_config(name), load_model(name)
prob, sp, _ = make_problem(cfg, model, num_steps=N)
sp.scaling = False; sp.equality_constraints = False
q = synthetic_trajectory(cfg, model, N, seed=0, lower=0.01)
dev = hip.HipPath(model, prob, sp)
dev.set_q(q)
dev.set_option("solver_nd", 0)
dev.gn_step(); p0 = dev.get("step").copy()
dev.set_option("solver_nd", 1)
dev.gn_step(); p1 = dev.get("step").copy()
print("status", dev.solver_status())
orc = Oracle(model, prob, sp)
g, bands = orc.grad_hess(q)
Hd = ol.penta_make_dense(*bands)
pref, unc = ol.refined_solution(Hd, -g.ravel())
pn = np.abs(pref).max()
print(name, N, "err two-sided %.3e  err nd %.3e  diff %.3e  finite %s" % (np.abs(p0 - pref).max() / pn, np.abs(p1 - pref).max() / pn, np.abs(p1 - p0).max() / pn, np.all(np.isfinite(p1))))
nq = model.nq
e = np.abs(p1 - pref).reshape(N + 1, nq).max(axis=1) / pn
print("per-row error:", np.array2string(e, precision=1, max_line_width=200))
