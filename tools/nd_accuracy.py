#!/usr/bin/env python3
"""Forward error of the solver variants against an extended-precision solution (tests/oracle_lib.py
refined_solution), several horizons and trajectory seeds: pipelined nested dissection (penta_pipe.h), the
seven-workgroup nested dissection (penta_nd.h), the two-workgroup LDL^T, and the reference's pivoted-LU block
Thomas (penta_kernel).  Prints error / LU-error ratios.  Usage: python tools/nd_accuracy.py [model] [N ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

import oracle_lib as ol  # noqa: E402
from idto_amd import hip  # noqa: E402
from idto_amd.model import load_model  # noqa: E402
from idto_amd.problem import load_config, make_problem, synthetic_trajectory  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "mini_cheetah"
Ns = [int(a) for a in sys.argv[2:]] or [24, 31, 40]
cfg, model = load_config(name), load_model(name)
for N in Ns:
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    sp.scaling = sp.equality_constraints = False
    orc = Oracle(model, prob, sp)
    for seed in range(4):
        q = synthetic_trajectory(cfg, model, N, seed=seed, lower=0.01 if name in ("hopper", "mini_cheetah") else 0.0)
        if name == "spinner":
            q[:, 1] = np.linspace(1.5, 1.25, N + 1)
        dev = hip.HipPath(model, prob, sp)
        dev.set_q(q)
        out = {}
        for label, opts in (("band", {"solver_band": 2}), ("pipe", {"solver_band": 0, "solver_pipe": 1}), ("ptail", {"debug_pipe_tail": 1}), ("nd", {"solver_pipe": 0, "debug_pipe_tail": 0}), ("two", {"solver_nd": 0}),
                            ("lu", {"reference_solver": 1})):
            for k, v in opts.items():
                dev.set_option(k, v)
            dev.gn_step()
            out[label] = dev.get("step")
            out[label + "_solver"] = dev.get_option("last_solver")
        dev.close()
        g, bands = orc.grad_hess(q)
        p_ref, unc = ol.refined_solution(ol.penta_make_dense(*bands), -g.ravel())
        pn = np.abs(p_ref).max()
        codes = {k[:-7]: out.pop(k) for k in [k for k in out if k.endswith("_solver")]}
        err = {k: np.abs(v.ravel() - p_ref).max() / pn for k, v in out.items()}
        gn = np.abs(g).max() + 1e-300
        res = {k: np.abs(ol.penta_multiply(*bands, v) + g).max() / gn for k, v in out.items()}
        ab = [np.abs(b) for b in bands]
        bwd = {k: (np.abs(ol.penta_multiply(*bands, v) + g.ravel()) / (ol.penta_multiply(*ab, np.abs(v)) + np.abs(g.ravel()) + 1e-300)).max() for k, v in out.items()}
        print(f"{name} N={N} seed={seed}: residual/|g| lu {res['lu']:.1e} pipe {res['pipe']:.1e} ptail {res['ptail']:.1e} nd {res['nd']:.1e} two {res['two']:.1e}")
        print(f"{name} N={N} seed={seed}: componentwise backward error lu {bwd['lu']:.1e} pipe {bwd['pipe']:.1e} ptail {bwd['ptail']:.1e} nd {bwd['nd']:.1e} two {bwd['two']:.1e}")
        if codes["band"] == 6:   # the scalar band factorisation (csrc/penta_band.h) took this block size
            print(f"{name} N={N} seed={seed}: band: residual/|g| {res['band']:.1e}, componentwise backward error {bwd['band']:.1e}, forward error band/lu {err['band'] / err['lu']:.2f}")
        print(f"{name} N={N} seed={seed}: lu {err['lu']:.2e}  pipe/lu {err['pipe'] / err['lu']:.2f}  nd/lu {err['nd'] / err['lu']:.2f}  "
              f"two/lu {err['two'] / err['lu']:.2f}  ptail/lu {err['ptail'] / err['lu']:.2f}  (unc {unc:.1e}; kernels that ran, "
              f"4 pipelined / 2 nested dissection / 1 two workgroups / 3 LU: pipe {codes['pipe']} ptail {codes['ptail']} nd {codes['nd']} two {codes['two']})", flush=True)
