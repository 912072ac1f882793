"""Stress for run-to-run determinism of the device optimizer (debugging aid)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem
from idto_amd.optimizer import TrajectoryOptimizer, TrajectoryOptimizerSolution, TrajectoryOptimizerStats
from oracle_lib import Oracle

def problem(name, N, method, ls, eq, scaling=False):
    cfg, model = load_config(name), load_model(name)
    prob, sp, q_guess = make_problem(cfg, model, num_steps=N)
    sp.max_iterations, sp.verbose, sp.num_threads = 6, False, 1
    sp.method, sp.linesearch_method = method, ls
    sp.scaling, sp.equality_constraints = scaling, eq
    return model, prob, sp, q_guess

def dev_costs(model, prob, sp, q_guess):
    opt = TrajectoryOptimizer(model, prob, sp)
    sol, st = TrajectoryOptimizerSolution(), TrajectoryOptimizerStats()
    opt.Solve(q_guess, sol, st)
    c = np.array(st.iteration_costs)
    opt.close()
    return c

cases = [("acrobot", 20, "linesearch", "backtracking", False), ("acrobot", 10, "trust_region", "armijo", True),
         ("hopper", 20, "linesearch", "backtracking", True), ("spinner", 20, "linesearch", "armijo", False)]
noise = [problem("mini_cheetah", 20, "linesearch", "armijo", False), problem("hopper", 20, "trust_region", "armijo", True, True)]
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    for nz in noise:
        dev_costs(*nz)
    for cs in cases:
        P = problem(*cs)
        ref = np.array(Oracle(P[0], P[1], P[2]).solve(P[3])["stats"].iteration_costs)
        c = dev_costs(*P)
        if not np.allclose(c, ref, rtol=1e-6):
            bad += 1
            c2 = dev_costs(*P)
            ref2 = np.array(Oracle(P[0], P[1], P[2]).solve(P[3])["stats"].iteration_costs)
            print("MISMATCH rep", rep, cs, "\n dev ", c, "\n dev2", c2, "\n ref ", ref, "\n ref2", ref2, flush=True)
print("mismatches:", bad)
