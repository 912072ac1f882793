"""Full trust-region iteration of the small models through idto_hip_tr_solve, with and without the one-workgroup launch
(option tr_small), constraints off / enforced: ms per iteration, median of 5 solves of 40 iterations."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, SCALING
iters = 40
for name in sys.argv[1:] or ["acrobot", "spinner"]:
    cfg, model = load_config(name), load_model(name)
    prob, sp, q_guess = make_problem(cfg, model, num_steps=40)
    for con in (False, True):
        for small, fold in ((1, 1), (1, 0), (0, 0)):
            dev = hip.HipPath(model, prob, sp)
            dev.set_option("tr_small", small)
            dev.set_option("tr_fold", fold)
            ts = []
            for _ in range(6):
                dev.set_q(np.asarray(q_guess).ravel())
                dev.eval_tau()
                t0 = time.perf_counter()
                rows, _ = dev.tr_solve(iters, SCALING["double_sqrt"], True, False, 1e-1, 1e5,
                                       constrained_dofs=model.unactuated_dofs if con else ())
                ts.append(time.perf_counter() - t0)
            print(f"{name} constraints {'enforced' if con else 'off'} tr_small={small} tr_fold={fold}: {1e3 * np.median(ts[1:]) / iters:.4f} ms/iteration "
                  f"(accepted {int(rows[:, 9].sum())}/{iters}, solver {dev.get_option('last_solver')})")
            dev.close()
