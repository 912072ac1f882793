"""In-kernel clock stamps of fd_kernel (a library built with -DIDTO_FD_STAMPS: tools/fd_variants.sh stamps "-DIDTO_FD_STAMPS";
run with IDTO_HIP_LIB=build/variants/stamps/libidto_hip.so).  Block k = 1, lanes tid 0 (the record's own evaluation,
path 0) and tid 192 (a mass-matrix column): shader-clock cycles since the block's start."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory
NAMES = ["start", "loads issued", "barrier 1", "v/a/edq + barrier 2 + outputs issued", "eval: inputs in registers", "eval: sincos",
         "eval: common body + its pairs", "eval: slot 0 done", "eval: last slot kinematics", "eval: last slot pairs",
         "eval: trailing pairs", "eval: backward pass", "evaluations + barrier", "record", "products", "barrier"]
for name, N in ((sys.argv[1], int(sys.argv[2])),) if len(sys.argv) > 2 else (("mini_cheetah", 40), ("allegro_hand", 60)):
    cfg = load_config(name); model = load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    q = synthetic_trajectory(cfg, model, N, seed=0, lower=0.01)
    dev = hip.HipPath(model, prob, sp); dev.set_q(q)
    acc = np.zeros(32); n = 0
    for it in range(30):
        dev.eval_partials(); dev.sync()
        v = np.zeros(dev.array_size("nplus")); hip._chk(hip.lib().idto_hip_get(dev.h, hip.ARR["nplus"], hip.dptr(v))); v = v[2 * model.nq * model.nv:][:32]
        if it >= 10: acc += v; n += 1
    acc /= n
    print(name, "N =", N, "(cycles since the block's start; 100 MHz ticks if the counter is the constant one)")
    for i, nm in enumerate(NAMES):
        print(f"  {i:2d} {nm:42s} tid0 {acc[i]:9.0f} (+{acc[i] - (acc[i-1] if i else 0):7.0f})   tid192 {acc[16+i]:9.0f}")
    dev.close()
