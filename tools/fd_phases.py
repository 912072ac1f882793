"""fd_kernel timing: mode 0 (tau only: ONE evaluation per block, one wavefront) vs mode 1
(57 evaluations per block, four wavefronts) - separates the latency of a single inverse-dynamics
evaluation (+ prologue) from the cost of the batch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory
for name, N in (("mini_cheetah", 40), ("hopper", 50), ("allegro_hand", 60)):
    cfg = load_config(name); model = load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    q = synthetic_trajectory(cfg, model, N, seed=0, lower=0.01)
    dev = hip.HipPath(model, prob, sp)
    dev.set_q(q)
    for label, fn in (("eval_tau (1 evaluation/block + cost kernel)", dev.eval_tau), ("eval_partials", dev.eval_partials)):
        for _ in range(20): fn()
        dev.sync(); t0 = time.perf_counter()
        for _ in range(300): fn()
        dev.sync()
        print(f"{name:14s} {label:46s} {1e6 * (time.perf_counter() - t0) / 300:7.1f} us")
    dev.close()

print("fd_kernel (mini_cheetah N=40) truncated after: 1 N+/v/a, 2 evaluation inputs, 3 inverse dynamics, 0 complete")
cfg = load_config("mini_cheetah"); model = load_model("mini_cheetah")
prob, sp, _ = make_problem(cfg, model, num_steps=40)
q = synthetic_trajectory(cfg, model, 40, seed=0, lower=0.01)
dev = hip.HipPath(model, prob, sp); dev.set_q(q)
for stop in (1, 2, 3, 0):
    dev.set_option("fd_stop", stop)
    for _ in range(20): dev.eval_partials()
    dev.sync(); dev.timing_enable(True); dev.timing_reset()
    for _ in range(200): dev.eval_partials()
    dev.sync()
    print(f"  fd_stop={stop}: {1e3 * dev.timing_get(0)[0]:.2f} us")
    dev.timing_enable(False)
