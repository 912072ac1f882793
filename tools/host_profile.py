"""Wall-clock profile of the host-side trust-region loop (libidto_opt.so) per phase:
IDTO_OPT_PROFILE=1 python tools/host_profile.py [config] [N] [iterations]"""
import os, sys
os.environ["IDTO_OPT_PROFILE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem
from idto_amd.optimizer import TrajectoryOptimizer, TrajectoryOptimizerSolution, TrajectoryOptimizerStats

name = sys.argv[1] if len(sys.argv) > 1 else "mini_cheetah"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cfg = load_config(name); model = load_model(name)
prob, sp, q_guess = make_problem(cfg, model, num_steps=N)
sp.max_iterations, sp.verbose = iters, False
opt = TrajectoryOptimizer(model, prob, sp)
for rep in range(2):  # the profile covers the second (warmed-up) solve only
    sol, st = TrajectoryOptimizerSolution(), TrajectoryOptimizerStats()
    opt.Solve(q_guess, sol, st)
    print(f"{name} N={N} n_eq={opt.num_equality_constraints()}: {1e3 * st.solve_time / iters:.3f} ms/iteration "
          f"({iters} iterations, solve {rep})", flush=True)
opt.close()
