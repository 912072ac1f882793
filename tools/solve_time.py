"""Per-iteration wall time of the full trust-region iteration (scaling + equality constraints as
in the reference's example YAMLs): host-side TrajectoryOptimizer on the MI355X vs the CPU oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from idto_amd.model import load_model
from idto_amd.optimizer import TrajectoryOptimizer, TrajectoryOptimizerSolution, TrajectoryOptimizerStats
from idto_amd.problem import load_config, make_problem
from oracle_lib import Oracle
for name in ("spinner", "hopper", "mini_cheetah", "allegro_hand"):
    cfg = load_config(name); model = load_model(name)
    prob, sp, q_guess = make_problem(cfg, model)
    sp.max_iterations, sp.verbose = 20, False
    for nt in (1, 4):
        sp.num_threads = nt
        r = Oracle(model, prob, sp).solve(q_guess)
        print(f"{name:13s} N={prob.num_steps:3d} oracle {nt} thread(s): {1e3 * r['stats'].solve_time / 20:8.2f} ms/iteration")
    opt = TrajectoryOptimizer(model, prob, sp)
    sol, st = TrajectoryOptimizerSolution(), TrajectoryOptimizerStats()
    opt.Solve(q_guess, sol, st)
    sol, st = TrajectoryOptimizerSolution(), TrajectoryOptimizerStats()
    t0 = time.perf_counter(); opt.Solve(q_guess, sol, st); dt = time.perf_counter() - t0
    print(f"{name:13s} N={prob.num_steps:3d} MI355X host loop:    {1e3 * dt / 20:8.2f} ms/iteration (n_eq = {opt.num_equality_constraints()})")
