#!/usr/bin/env python3
"""One leg of bench.py's `cpu_baseline`: times Gauss-Newton steps of the CPU oracle (the port of the
reference algorithm, OpenMP where the reference has `#pragma omp parallel for`: TO.cc:209, :476) at ONE
num_threads value, in a process of its own so that the OpenMP runtime starts with the placement this leg
asks for (bench.py's own process has torch's OpenMP runtime loaded and an environment that is already
read).  Prints one JSON line.  Test / measurement infrastructure only: nothing under idto_amd/ uses it.

  python tools/cpu_baseline.py --config mini_cheetah --num-steps 40 --threads 8 --iters 300 --repeats 3

The caller sets OMP_PROC_BIND / OMP_PLACES / OMP_WAIT_POLICY in the environment (bench.py: close / cores
/ active); they are echoed in the output.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="mini_cheetah")
    ap.add_argument("--num-steps", type=int, default=40)
    ap.add_argument("--threads", type=int, default=1)
    ap.add_argument("--iters", type=int, default=0, help="timed iterations per repeat (fixed by the caller: every leg and "
                    "every run times the same sample); 0: derive from --budget")
    ap.add_argument("--repeats", type=int, default=3, help="repeats of the timed sample; the median is the leg's rate")
    ap.add_argument("--budget", type=float, default=4.0, help="seconds of timed work when --iters is 0")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()

    from idto_amd.model import load_model
    from idto_amd.problem import load_config, make_problem, synthetic_trajectory
    from oracle_lib import Oracle

    cfg = load_config(args.config)
    model = load_model(args.config)
    prob, sp, _ = make_problem(cfg, model, num_steps=args.num_steps)
    sp.scaling = False
    sp.equality_constraints = False
    sp.num_threads = args.threads
    q = synthetic_trajectory(cfg, model, args.num_steps, seed=args.seed, lower=0.01)
    orc = Oracle(model, prob, sp)
    orc.time_gn_steps(q, 2)                           # warm-up (thread pool, caches)
    t1 = orc.time_gn_steps(q, 5)
    iters = args.iters if args.iters > 0 else max(5, min(2000, int(args.budget / t1)))
    ts = sorted(orc.time_gn_steps(q, iters) for _ in range(max(1, args.repeats)))
    t = ts[len(ts) // 2]
    # where one step's time goes: the two OpenMP loops (tau, finite differences) against the serial
    # rest (N+, v, a, assembly, factor + solve) -- the Amdahl ceiling of the reference's parallelisation
    parts = orc.time_gn_parts(q, max(3, iters // 4)) if hasattr(orc, "time_gn_parts") else None
    out = {"num_threads": args.threads, "iters": iters, "repeats": len(ts), "s_per_iter": t, "iters_per_s": 1.0 / t,
           "iters_per_s_repeats": [1.0 / x for x in ts], "spread": (ts[-1] - ts[0]) / t,
           "omp": {k: os.environ.get(k) for k in ("OMP_PROC_BIND", "OMP_PLACES", "OMP_WAIT_POLICY", "OMP_DYNAMIC")},
           "affinity_cores": len(os.sched_getaffinity(0))}
    if parts:
        out["parts_s"] = parts
    print(json.dumps(out))


if __name__ == "__main__":
    main()
