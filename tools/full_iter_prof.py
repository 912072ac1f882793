"""One warmed-up TrajectoryOptimizer::Solve of the mini_cheetah example (for rocprofv3 --kernel-trace --stats)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem
from idto_amd.optimizer import TrajectoryOptimizer, TrajectoryOptimizerSolution, TrajectoryOptimizerStats
name = sys.argv[1] if len(sys.argv) > 1 else "mini_cheetah"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg, model = load_config(name), load_model(name)
prob, sp, q_guess = make_problem(cfg, model, num_steps=40 if name != "allegro_hand" else 60)
sp.max_iterations, sp.verbose = iters, False
opt = TrajectoryOptimizer(model, prob, sp)
for _ in range(3):
    sol, st = TrajectoryOptimizerSolution(), TrajectoryOptimizerStats()
    t0 = time.perf_counter()
    opt.Solve(q_guess, sol, st)
    dt = time.perf_counter() - t0
print(f"{name}: {1e3 * st.solve_time / iters:.4f} ms/iteration (wall {1e3 * dt / iters:.4f}), rho>0 in {int((st.trust_ratios > 0).sum())}/{iters}")
