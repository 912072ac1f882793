"""fd_kernel by truncation (option fd_stop): HIP-event time of 10 the launch alone, after 8 the loads, 9 N+, 1 v/a, 2 the evaluation inputs, 3 the inverse
dynamics, 4 the record, 5 (barrier), 6 the lower-triangle products, 7 all products, 0 complete."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory
for name, N in (("mini_cheetah", 40), ("allegro_hand", 60), ("hopper", 50)):
    cfg = load_config(name); model = load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    q = synthetic_trajectory(cfg, model, N, seed=0, lower=0.01)
    dev = hip.HipPath(model, prob, sp); dev.set_q(q)
    for fast in ((0, 1) if "--both" in sys.argv else (1,)):
        dev.set_option("fd_fast", fast)
        out = []
        for stop in (10, 8, 9, 1, 2, 3, 4, 6, 7, 0):
            dev.set_option("fd_stop", stop)
            for _ in range(20): dev.eval_partials()
            dev.sync(); dev.timing_enable(True); dev.timing_reset()
            for _ in range(200): dev.eval_partials()
            dev.sync()
            out.append(f"{stop}: {1e3 * dev.timing_get(0)[0]:.2f}")
            dev.timing_enable(False)
        dev.set_option("fd_stop", 0)
        print(f"{name} N={N} fd_fast={fast}  fd_stop -> us  " + "  ".join(out), flush=True)
    dev.close()
