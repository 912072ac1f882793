"""runs fd_kernel alone (eval_partials) with a given fd_stop: the workload of tools/fd_pmc.sh"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory
name, N, stop = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
fast = int(sys.argv[4]) if len(sys.argv) > 4 else 1
cfg = load_config(name); model = load_model(name)
prob, sp, _ = make_problem(cfg, model, num_steps=N)
q = synthetic_trajectory(cfg, model, N, seed=0, lower=0.01)
dev = hip.HipPath(model, prob, sp); dev.set_q(q)
dev.set_option("fd_fast", fast)
dev.set_option("fd_stop", stop)
for _ in range(30): dev.eval_partials()
dev.sync()
dev.close()
