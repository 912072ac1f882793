"""Time of the multi-right-hand-side device solve used by CalcLagrangeMultipliers (host arrays in/out)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory
for name, N, nrhs in (("mini_cheetah", 20, 121), ("allegro_hand", 40, 241), ("hopper", 40, 121)):
    cfg = load_config(name); model = load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    q = synthetic_trajectory(cfg, model, N, seed=0, lower=0.01)
    dev = hip.HipPath(model, prob, sp); dev.set_q(q); dev.eval_partials(); dev.grad_hess(); dev.sync()
    rhs = np.random.default_rng(0).normal(size=(nrhs, (N + 1) * model.nq))
    for _ in range(3): dev.solve_host(rhs)
    t0 = time.perf_counter()
    for _ in range(10): dev.solve_host(rhs)
    t = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(10): dev.get("H_A"); 
    tg = (time.perf_counter() - t0) / 10
    print(f"{name:13s} N={N} nrhs={nrhs}: solve_host {1e6 * t:8.1f} us; one idto_hip_get(H_A) {1e6 * tg:6.1f} us")
