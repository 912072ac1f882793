#!/usr/bin/env python3
"""Bit reproducibility of the constrained trust-region loop (banded KKT step, csrc/kkt.h) over fresh contexts: every
solve of the same problem must give the same rows and iterate, with clean flags - a race between the workgroups of
the factorisation would show as a difference.  Usage: python tools/stress_kkt.py [model N repetitions]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from idto_amd import hip  # noqa: E402
from idto_amd.model import load_model  # noqa: E402
from idto_amd.problem import load_config, make_problem  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "allegro_hand"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
cfg, model = load_config(name), load_model(name)
prob, sp, q0 = make_problem(cfg, model, num_steps=N)
dofs = np.asarray(model.unactuated_dofs)
ref = None
bad = 0
t0 = time.perf_counter()
for rep in range(reps):
    dev = hip.HipPath(model, prob, sp)
    for inner in range(3):
        dev.set_q(np.asarray(q0))
        dev.eval_tau()
        rows, delta = dev.tr_solve(6, 2, True, False, 1e-1, 1e5, constrained_dofs=dofs)
        cols = [c for c in range(17) if c != 10]
        out = (rows[:, cols].copy(), dev.get("q").copy(), dev.get("con_lambda").copy())
        if ref is None:
            ref = out
        same = all(np.array_equal(a, b) for a, b in zip(out, ref))
        clean = bool((rows[:, 14] == 0).all())
        if not (same and clean):
            bad += 1
            print(f"rep {rep}.{inner}: same {same} clean {clean} flags {sorted(set(rows[:, 14].astype(int)))}", flush=True)
    solver = dev.get_option("kkt_last_solver")
    dev.close()
print(f"{name} N={N}: {3 * reps} constrained solves of 6 iterations over {reps} fresh contexts (KKT factorisation: kernel {solver}), "
      f"{bad} differing or flagged, {time.perf_counter() - t0:.1f} s")
sys.exit(1 if bad else 0)
