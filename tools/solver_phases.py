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
dall = dev.get("debug")
n = N + 1
names = ["stage+sync", "products+sync", "load cols", "eliminate", "store+sync -> next row"]
for w in range(4):
    d = dall[w * (n + 3) * 8:(w + 1) * (n + 3) * 8].reshape(-1, 8)
    rows = d[:n]
    if w == 0:
        ph = [rows[:, 1] - rows[:, 0], rows[:, 2] - rows[:, 1], rows[:, 3] - rows[:, 2], rows[:, 4] - rows[:, 3]]
        ph.append(np.concatenate([rows[1:, 0] - rows[:-1, 4], [d[n, 0] - rows[-1, 4]]]))
        tot_fwd = d[n, 0] - rows[0, 0]
        bwd = d[n, 1] - d[n, 0]
        print(f"{name} N={N}: wave 0 cycles per block row (median over rows 2..n-1)")
        for nm, p in zip(names, ph):
            print(f"  {nm:28s} {np.median(p[2:]):10.0f}")
        print(f"  forward total {tot_fwd:.0f} cycles, backward total {bwd:.0f} cycles ({bwd / (n - 1):.0f}/row)")
        t0 = rows[:, 0]
    else:
        print(f"  wave {w}: stage+sync {np.median((rows[:,1]-rows[:,0])[2:]):8.0f}  products+sync {np.median((rows[:,2]-rows[:,1])[2:]):8.0f}"
              f"  row-to-row {np.median(np.diff(rows[:,0])[2:]):8.0f}  start skew vs wave0 {np.median((rows[:,0]-t0)[2:]):8.0f}")

print("arrival at barriers relative to row start (median cycles): [before b1, after b1, before b2, after b2, before b3], next row start")
for w in range(4):
    d = dall[w * (n + 3) * 8:(w + 1) * (n + 3) * 8].reshape(-1, 8)
    rows = d[:n]
    base = dall[0:(n + 3) * 8].reshape(-1, 8)[:n, 0]
    rel = lambda col: np.median((rows[:, col] - base)[2:-1])
    nxt = np.median((rows[1:, 0] - base[:-1])[2:])
    print(f"  wave {w}: {rel(7):7.0f} {rel(1):7.0f} {rel(5):7.0f} {rel(2):7.0f} {rel(6):7.0f}   {nxt:7.0f}")

print("helper I/O waves: [after stage, after fetch, before b3] relative to after-b2")
for w in (2, 3):
    d = dall[w * (n + 3) * 8:(w + 1) * (n + 3) * 8].reshape(-1, 8)
    rows = d[:n]
    f = lambda a, b: np.median((rows[:, a] - rows[:, b])[2:-1])
    print(f"  wave {w}: {f(3, 2):7.0f} {f(4, 2):7.0f} {f(6, 2):7.0f}")
