#!/usr/bin/env python3
"""A wider sweep than tests/test_gpu_fast_shape.py: the straight-line evaluation (id_fast.h) against the generic one, bit for
bit, over many trajectory seeds, horizons and all three derivative modes, on every model with an instantiated shape; the
generic one against the oracle on a subset.  Usage: python tools/fd_sweep.py [seeds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

from idto_amd import hip  # noqa: E402
from idto_amd.model import load_model  # noqa: E402
from idto_amd.problem import load_config, make_problem, synthetic_trajectory  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

KEYS = ("v", "a", "tau", "dtau_dqp", "dtau_dqt", "dtau_dqm", "gradient", "H_A", "H_B", "H_C")


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
t0 = time.perf_counter()
cases = bad = oracle_cases = 0
for name, horizons, lowers in (("acrobot", (3, 17, 40), (0.0,)), ("spinner", (4, 40), (0.0,)), ("hopper", (5, 50), (0.01, 0.05)),
                               ("mini_cheetah", (2, 13, 40), (0.0, 0.01, 0.03)), ("allegro_hand", (3, 60), (0.0,))):
    cfg, model = load_config(name), load_model(name)
    for N in horizons:
        prob, sp, _ = make_problem(cfg, model, num_steps=N)
        sp.scaling = sp.equality_constraints = False
        devs = {}
        for fast in (1, 0):
            devs[fast] = hip.HipPath(model, prob, sp)
            devs[fast].set_option("fd_fast", fast)
        orc = Oracle(model, prob, sp)
        for seed in range(seeds):
            for lower in lowers:
                q = synthetic_trajectory(cfg, model, N, seed=100 + seed, lower=lower)
                if name == "spinner" and seed % 2:
                    q[:, 1] = np.linspace(1.5, 1.25, N + 1)   # (finger tip on the spinner)
                for method in ((0, 1, 2) if seed < 2 else (0,)):
                    out = {}
                    for fast in (1, 0):
                        d = devs[fast]
                        d.set_option("gradients_method", method)
                        d.set_q(q)
                        d.eval_partials()
                        d.grad_hess()
                        out[fast] = {k: d.get(k) for k in KEYS}
                    cases += 1
                    diff = [k for k in KEYS if not same(out[0][k], out[1][k])]
                    if diff:
                        bad += 1
                        print(f"{name} N={N} seed={seed} lower={lower} method={method}: fast != generic in {diff}", flush=True)
                    if method == 0 and seed < 3:
                        g, bands = orc.grad_hess(q)
                        oracle_cases += 1
                        if not (same(out[1]["gradient"], g) and same(out[1]["H_C"], bands[2])):
                            bad += 1
                            print(f"{name} N={N} seed={seed} lower={lower}: device != oracle", flush=True)
        for d in devs.values():
            d.close()
print(f"{cases} cases fast == generic ({oracle_cases} of them also == the oracle), {bad} differing, {time.perf_counter() - t0:.1f} s")
sys.exit(1 if bad else 0)
