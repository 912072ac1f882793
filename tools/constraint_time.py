"""Device time of the equality-constraint step (idto_hip_constraint_schur / _step) and of its
parts, wall clock around synchronising calls; run under rocprofv3 --kernel-trace --stats for the
per-kernel split."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory

cases = [(sys.argv[1], int(sys.argv[2]))] if len(sys.argv) > 2 else [("allegro_hand", 40), ("hopper", 40), ("mini_cheetah", 40)]
for name, N in cases:
    cfg = load_config(name); model = load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    q = synthetic_trajectory(cfg, model, N, seed=0, lower=0.01)
    dev = hip.HipPath(model, prob, sp); dev.set_q(q); dev.eval_partials(); dev.grad_hess(); dev.sync()
    dofs = [j for j in range(model.nv) if not model.actuated[j]] or list(range(min(6, model.nv)))
    neq = len(dofs) * N
    def timeit(f, reps=10):
        for _ in range(3): f()
        t0 = time.perf_counter()
        for _ in range(reps): f()
        return 1e6 * (time.perf_counter() - t0) / reps
    t_schur = timeit(lambda: (dev.grad_hess(), dev.constraint_schur(dofs)))  # (grad_hess: a new Hessian each time)
    S, Jy = dev.constraint_schur(dofs)
    lam = np.linalg.solve(S + 1e-9 * np.eye(neq), -Jy)
    t_step = timeit(lambda: dev.constraint_step(lam))
    t_one = timeit(lambda: (dev.factor_solve(), dev.sync()))
    import torch
    n = (N + 1) * model.nq
    rhs = torch.randn(neq + 1, n, dtype=torch.float64, device="cuda"); x = torch.zeros_like(rhs)
    t_multi = timeit(lambda: (dev.factor_solve(rhs.data_ptr(), neq + 1, x.data_ptr()), dev.sync()))
    t_two = timeit(lambda: (dev.factor_solve(rhs.data_ptr(), 2, x.data_ptr()), dev.sync()))
    print(f"{name:13s} N={N} n_eq={neq}: grad_hess + constraint_schur {t_schur:8.1f} us, constraint_step {t_step:6.1f} us, "
          f"single-rhs factor_solve {t_one:6.1f} us, {neq + 1}-rhs factor_solve (device buffers) {t_multi:6.1f} us, "
          f"2-rhs {t_two:6.1f} us")
    dev.close()
