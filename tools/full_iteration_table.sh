#!/bin/bash
# Full trust-region iteration (TrajectoryOptimizer::Solve through libidto_opt.so, example YAML
# settings) per config with the host-side phase profile, the CPU oracle beside it, and the device
# time of the equality-constraint step.  Output: gpurun_out/full_iteration.txt
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for c in "acrobot 40" "spinner 40" "hopper 50" "mini_cheetah 40" "allegro_hand 60"; do
  IDTO_OPT_PROFILE=1 timeout 120 python tools/host_profile.py $c 20 2>&1 | grep -v amdgpu.ids
  timeout 300 python - $c <<'PY'
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem
from oracle_lib import Oracle
name, N = sys.argv[1], int(sys.argv[2])
cfg = load_config(name); model = load_model(name)
prob, sp, q_guess = make_problem(cfg, model, num_steps=N)
iters = 20 if name != "allegro_hand" else 6
sp.max_iterations, sp.verbose = iters, False
sp.num_threads = 1
if name != "allegro_hand":
    Oracle(model, prob, sp).solve(q_guess)  # warm-up (page-in, OpenMP pool)
for nt in (1, 4):
    sp.num_threads = nt
    t0 = time.perf_counter(); Oracle(model, prob, sp).solve(q_guess)
    print(f"{name} N={N} CPU oracle, num_threads={nt}: {1e3 * (time.perf_counter() - t0) / iters:.3f} ms/iteration")
PY
done
timeout 100 python tools/constraint_time.py 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee gpurun_out/full_iteration.txt
