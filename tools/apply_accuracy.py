import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory
for name, N in (("hopper", 50), ("mini_cheetah", 40), ("allegro_hand", 60), ("acrobot", 40), ("spinner", 40)):
    cfg, model = load_config(name), load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    sp.scaling = False; sp.equality_constraints = False
    q = synthetic_trajectory(cfg, model, N, seed=0, lower=0.01)
    dev = hip.HipPath(model, prob, sp)
    dev.set_option("solver_nd", 0)
    dev.set_q(q); dev.gn_step()
    g = dev.get("gradient").ravel()
    bands = [dev.get(k) for k in ("H_A", "H_B", "H_C")]
    Cs, Dm, Em = ol.penta_make_symmetric(*bands)
    Hd = ol.penta_make_dense(bands[0], bands[1], Cs, Dm, Em)
    ref, unc = ol.refined_solution(Hd, -g)
    x1 = dev.get("step").ravel()                      # in-kernel substitution (two-workgroup kernel)
    X = dev.solve_host(np.stack([-g, -g]))            # both columns through penta_apply_kernel
    sc = np.abs(ref).max()
    print(name, N, "err in-kernel %.2e  apply col0 %.2e col1 %.2e  unc %.1e" % (np.abs(x1 - ref).max() / sc, np.abs(X[0] - ref).max() / sc, np.abs(X[1] - ref).max() / sc, unc))
    dev.close()
