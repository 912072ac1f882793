"""Device time of idto_hip_constraint_solve (schur + dense LDL^T + step) per config; run under
rocprofv3 --kernel-trace --stats for the per-kernel split."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory

cases = [(sys.argv[1], int(sys.argv[2]))] if len(sys.argv) > 2 else [("allegro_hand", 60), ("hopper", 50), ("spinner", 40)]
for name, N in cases:
    cfg = load_config(name); model = load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    q = synthetic_trajectory(cfg, model, N, seed=0, lower=0.01)
    dev = hip.HipPath(model, prob, sp); dev.set_q(q); dev.eval_partials(); dev.sync()
    dofs = [j for j in range(model.nv) if not model.actuated[j]] or list(range(min(6, model.nv)))
    h = np.zeros(len(dofs) * N)
    def one():
        dev.grad_hess()
        return dev.constraint_solve(dofs, h)
    for _ in range(3): one()
    t0 = time.perf_counter()
    for _ in range(10): ok = one()[0]
    t = 1e6 * (time.perf_counter() - t0) / 10
    print(f"{name:13s} N={N} n_eq={h.size}: grad_hess + constraint_solve {t:8.1f} us (device factorisation used: {ok})")
    dev.close()
