import sys, os, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory
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
dev.set_option("solver_debug", 1)
for it in range(int(sys.argv[3]) if len(sys.argv) > 3 else 6):
    dev.factor_solve()
    st = dev.solver_status()
    try:
        p1 = dev.get("step").copy()
    except Exception as e:
        import ctypes as C
        out = np.zeros(dev.array_size("step")); 
        hip.lib().idto_hip_get(dev.h, hip.ARR["step"], hip.dptr(out)); p1 = out
    e = np.abs(p1 - p0).reshape(N + 1, model.nq).max(axis=1) / np.abs(p0).max()
    if it < 2 or not (e.max() < 1e-3) or st[0]:
        print(it, st, "max diff %.2e" % e.max(), "bad rows:", np.where(~(e < 1e-3))[0][:6])
print("done")
