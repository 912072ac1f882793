"""Phase timing of assemble_diag_kernel by truncation (option asm_stop): average kernel time when
the kernel returns after 1 staging, 2 N+/v, 3 weighted operands, 0 complete."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory
name, N = "mini_cheetah", 40
cfg = load_config(name); model = load_model(name)
prob, sp, _ = make_problem(cfg, model, num_steps=N)
q = synthetic_trajectory(cfg, model, N, seed=0, lower=0.01)
dev = hip.HipPath(model, prob, sp)
dev.set_q(q); dev.eval_partials(); dev.sync()
for stop in (1, 2, 3, 0):
    dev.set_option("asm_stop", stop)
    for _ in range(20):
        dev.grad_hess()
    dev.sync(); dev.timing_enable(True); dev.timing_reset()
    for _ in range(200):
        dev.grad_hess()
    dev.sync()
    print(f"asm_stop={stop}: {1e3 * dev.timing_get(1)[0]:.2f} us")
    dev.timing_enable(False)
