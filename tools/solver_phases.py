"""Per-phase cycle stamps of the block penta-diagonal solver kernel (option solver_debug).
Stamp layout (penta_ldl.h): dbg[((side * 4 + wave) * (n + 3) + local_row) * 8 + phase];
the entry after a side's last forward row holds [end of forward, end of backward]."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory

name, N = (sys.argv[1] if len(sys.argv) > 1 else "mini_cheetah"), int(sys.argv[2]) if len(sys.argv) > 2 else 40
two = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = load_config(name); model = load_model(name)
prob, sp, _ = make_problem(cfg, model, num_steps=N)
q = synthetic_trajectory(cfg, model, N, seed=0, lower=0.01)
dev = hip.HipPath(model, prob, sp)
dev.set_q(q); dev.set_option("solver_debug", 1); dev.set_option("two_sided", two)
dev.set_option("solver_nd", 0)   # (the stamps below are those of the one- / two-workgroup kernel)
for _ in range(3):
    dev.gn_step()
dev.sync()
dall = dev.get("debug")
n = N  # the solver skips the decoupled block row 0 of an assembled Hessian (idto_hip.hip SolverFirstRow)
m = (n - 1) // 2 if (two and n >= 10) else 0
sides = [(0, m + 2), (1, n - m - 2 + 2)] if m else [(0, n)]   # (side, forward rows incl. pseudo-rows)
names = ["stage+sync", "products+sync", "load cols", "eliminate", "store+sync -> next row"]
origin = None
for side, nf in sides:
    W = [dall[(side * 4 + w) * (n + 3) * 8:(side * 4 + w + 1) * (n + 3) * 8].reshape(-1, 8) for w in range(4)]
    d = W[0]
    rows = d[:nf]
    if origin is None:
        origin = rows[0, 0]
    ph = [rows[:, 1] - rows[:, 0], rows[:, 2] - rows[:, 1], rows[:, 3] - rows[:, 2], rows[:, 4] - rows[:, 3]]
    ph.append(np.concatenate([rows[1:, 0] - rows[:-1, 4], [d[nf, 0] - rows[-1, 4]]]))
    print(f"{name} N={N} side {side}: {nf} forward rows; wave 0 cycles per block row (median over rows 2..)")
    for nm, p in zip(names, ph):
        print(f"  {nm:28s} {np.median(p[2:nf - 3]):10.0f}")
    print(f"  start {rows[0, 0] - origin:.0f}, forward end {d[nf, 0] - origin:.0f}, backward end {d[nf, 1] - origin:.0f} "
          f"(forward {d[nf, 0] - rows[0, 0]:.0f}, backward {d[nf, 1] - d[nf, 0]:.0f}) cycles")
    if side == 0 and m:
        print(f"  join rows: row m took {rows[m + 1, 0] - rows[m, 0]:.0f}, row m+1 {d[nf, 0] - rows[m + 1, 0]:.0f} cycles")
    print("  arrival at barriers relative to row start: [before b1, after b1, before b2, after b2, before b3], next row start")
    base = rows[:, 0]
    for w in range(4):
        r = W[w][:nf]
        rel = lambda col: np.median((r[:, col] - base)[2:nf - 3])
        nxt = np.median((r[1:, 0] - base[:-1])[2:nf - 3])
        print(f"    wave {w}: {rel(7):7.0f} {rel(1):7.0f} {rel(5):7.0f} {rel(2):7.0f} {rel(6):7.0f}   {nxt:7.0f}")
