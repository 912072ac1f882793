"""Timeline of the fused Gauss-Newton launch (option "fused_debug"): per role, when its workgroups
start, see their inputs, and finish (us relative to the first workgroup's start)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory

name, N = (sys.argv[1] if len(sys.argv) > 1 else "mini_cheetah"), int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg, model = load_config(name), load_model(name)
prob, sp, _ = make_problem(cfg, model, num_steps=N)
sp.scaling = False; sp.equality_constraints = False
dev = hip.HipPath(model, prob, sp)
dev.set_q(synthetic_trajectory(cfg, model, N, seed=0, lower=0.01))
for _ in range(5):
    dev.gn_step()
dev.set_option("fused_debug", 1)
dev.gn_step(); dev.gn_step()
d = dev.get("debug")
nfd, nasm = N, 4 * (N + 1)
nb = nfd + nasm + 2
t = d[:4 * nb].reshape(nb, 4)[:, :3] / 100.0   # us
t0 = t[:, 0].min()
t = t - t0
for role, sl in (("fd", slice(0, nfd)), ("assemble", slice(nfd, nfd + nasm)), ("solver", slice(nfd + nasm, nb))):
    x = t[sl]
    print(f"{role:9s} start {x[:,0].min():7.2f}..{x[:,0].max():7.2f}  ready/body-done {x[:,1].min():7.2f}..{x[:,1].max():7.2f}"
          f"  end {x[:,2].min():7.2f}..{x[:,2].max():7.2f}")
asm = t[nfd:nfd + nasm]
for part in range(4):
    x = asm[part::4]
    dur = x[:, 2] - x[:, 1]
    print(f"assemble part {part}: body+signal {dur.min():5.2f}..{dur.max():5.2f} us (median {np.median(dur):5.2f}); rows of the slowest: {np.argsort(-dur)[:4]}")
