"""fd_kernel with the straight-line evaluation of csrc/id_fast.h (option fd_fast=1) against the generic
id_eval<MAXC> (fd_fast=0): bit comparison of every output (device against device, as int64 patterns, and
against the CPU oracle) and HIP-event kernel times, for the five BASELINE configurations at their horizons."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory

CASES = [("acrobot", 40, 0.0), ("spinner", 40, 0.0), ("hopper", 50, 0.01), ("mini_cheetah", 40, 0.01), ("allegro_hand", 60, 0.0)]
with_oracle = "--oracle" in sys.argv
if with_oracle:
    from oracle_lib import Oracle


def same(a, b):
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


for name, N, lower in CASES:
    cfg = load_config(name); model = load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    sp.scaling = False; sp.equality_constraints = False
    for seed in (0, 1):
        q = synthetic_trajectory(cfg, model, N, seed=seed, lower=lower)
        if name == "spinner":
            q[:, 1] = np.linspace(1.5, 1.25, N + 1)
        res = {}
        for fast in (0, 1):
            dev = hip.HipPath(model, prob, sp)
            dev.set_option("fd_fast", fast)
            dev.set_q(q)
            dev.eval_partials()
            dev.grad_hess()
            out = {k: dev.get(k) for k in ("tau", "dtau_dqp", "dtau_dqt", "dtau_dqm", "gradient", "H_A", "H_B", "H_C", "v", "a")}
            dev.eval_tau()
            out["tau0"] = dev.get("tau")
            if seed == 0:
                for _ in range(20): dev.eval_partials()
                dev.sync(); dev.timing_enable(True); dev.timing_reset()
                for _ in range(200): dev.eval_partials()
                dev.sync()
                out["t_fd"] = 1e3 * dev.timing_get(0)[0]
                dev.timing_enable(False)
                t0 = time.perf_counter()
                for _ in range(300): dev.gn_step()
                dev.sync()
                out["t_step"] = 1e6 * (time.perf_counter() - t0) / 300
            res[fast] = out
            dev.close()
        keys = [k for k in res[0] if not k.startswith("t_")]
        bad = [k for k in keys if not same(res[0][k], res[1][k])]
        signs = [k for k in keys if same(res[0][k], res[1][k]) and not np.array_equal(np.signbit(res[0][k]), np.signbit(res[1][k]))]
        line = f"{name:14s} N={N} seed={seed}: fast == generic: {'yes' if not bad else 'NO ' + str(bad)}"
        if signs: line += f" (signs of zeros differ in {signs})"
        if with_oracle:
            orc = Oracle(model, prob, sp)
            P = orc.eval_partials(q)
            tau = orc.eval_traj(q)[2]
            ob = [k for k in ("dtau_dqp", "dtau_dqt", "dtau_dqm") if not same(res[1][k], P[k])]
            if not same(res[1]["tau"], tau): ob.append("tau")
            line += f"; fast == oracle: {'yes' if not ob else 'NO ' + str(ob)}"
        if seed == 0:
            line += f"; fd_kernel {res[0]['t_fd']:.1f} -> {res[1]['t_fd']:.1f} us; gn_step {res[0]['t_step']:.1f} -> {res[1]['t_step']:.1f} us"
        print(line, flush=True)
