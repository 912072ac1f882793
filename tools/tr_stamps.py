"""Phase stamps of the trust-region kernels of the resident loop's last iteration (measurement build:
   bash tools/main_variants.sh trstamps -DIDTO_TR_STAMPS;  IDTO_HIP_LIB=build/variants/trstamps/libidto_hip.so python tools/tr_stamps.py [model] [iters])."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem
from idto_amd.hip import HipPath
name = sys.argv[1] if len(sys.argv) > 1 else "mini_cheetah"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cfg, model = load_config(name), load_model(name)
prob, sp, q_guess = make_problem(cfg, model, num_steps=40 if name != "allegro_hand" else 60)
d = HipPath(model, prob, sp)
fn = hip.lib().idto_hip_debug_tr_stamps
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
runs = []
for _ in range(5):
    d.set_q(np.asarray(q_guess).ravel())
    d.tr_solve(iters, 2, True, False, 1e-1, 1e5)
    buf = (ctypes.c_ulonglong * 64)()
    assert fn(buf) == 0
    runs.append(np.array(buf[:], dtype=np.float64) / 100.0)   # us
r = np.median(np.array(runs[1:]), axis=0)
t0 = r[0]
names = {0: "tr_iter entry (block 0)", 1: "D, g~, w of 5 block rows staged", 2: "band products", 3: "sums of the row published",
         4: "sums of all block rows polled", 5: "added in block order", 6: "convergence criteria", 7: "dogleg", 9: "trial point of the row, end",
         16: "cost entry", 17: "columns", 18: "terms", 19: "cost", 20: "decision", 21: "q <- q_trial"}
print(f"{name}: stamps of the last iteration, us from tr_iter_kernel's entry (median of 4 solves)")
for k in sorted(names):
    if r[k] > 0: print(f"  {names[k]:38s} {r[k] - t0:8.2f}")
print(f"  tr_iter entry -> cost entry (fd_kernel in between) {r[16] - r[0]:.2f}; cost entry -> end {r[21] - r[16]:.2f}")
