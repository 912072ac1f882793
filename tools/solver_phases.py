"""Per-phase cycle stamps of the block-Thomas solver kernel (option solver_debug)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory

name, N = (sys.argv[1] if len(sys.argv) > 1 else "mini_cheetah"), int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = load_config(name); model = load_model(name)
prob, sp, _ = make_problem(cfg, model, num_steps=N)
q = synthetic_trajectory(cfg, model, N, seed=0, lower=0.01)
dev = hip.HipPath(model, prob, sp)
dev.set_q(q); dev.set_option("solver_debug", 1)
for _ in range(3):
    dev.gn_step()
dev.sync()
d = dev.get("debug").reshape(-1, 8)
n = N + 1
rows = d[:n]
names = ["stage+sync", "products+sync", "load cols", "eliminate", "store+sync -> next row"]
ph = [rows[:, 1] - rows[:, 0], rows[:, 2] - rows[:, 1], rows[:, 3] - rows[:, 2], rows[:, 4] - rows[:, 3]]
ph.append(np.concatenate([rows[1:, 0] - rows[:-1, 4], [d[n, 0] - rows[-1, 4]]]))
tot_fwd = d[n, 0] - rows[0, 0]
bwd = d[n, 1] - d[n, 0]
print(f"{name} N={N}: cycles per block row (median over rows 2..n-1)")
for nm, p in zip(names, ph):
    print(f"  {nm:28s} {np.median(p[2:]):10.0f}")
print(f"  forward total {tot_fwd:.0f} cycles, backward total {bwd:.0f} cycles ({bwd / (n - 1):.0f}/row)")
