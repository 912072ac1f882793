"""Waits between the workgroups of the multi-workgroup solvers are bounded (csrc/penta_ldl.h spin_wait): a launch
whose partner workgroup never shows up ends by itself, reports IDTO_HIP_SOLVER_TIMEOUT, and the context steps down
to a variant with fewer co-resident workgroups - the device never hangs.  The option "debug_skip_role" makes one
role of the nested-dissection kernels return at once, which is what a partner that is not resident looks like."""
import numpy as np
import pytest

from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,N,role", [("mini_cheetah", 40, 0), ("hopper", 50, 1)])
def test_a_missing_partner_workgroup_times_out_and_the_solve_is_repeated(name, N, role):
    cfg, model = load_config(name), load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    sp.scaling = sp.equality_constraints = False
    q = synthetic_trajectory(cfg, model, N, seed=3, lower=0.01)
    ref = hip.HipPath(model, prob, sp)
    ref.set_option("solver_nd", 0)
    ref.set_q(q)
    ref.gn_step()
    want = ref.get("step")
    ref.close()
    dev = hip.HipPath(model, prob, sp)
    dev.set_q(q)
    dev.gn_step()
    assert dev.get_option("last_solver") in (2, 4)        # nested dissection (pipelined chains when the block size allows)
    first = dev.get("step")
    dev.set_option("debug_skip_role", role)               # a producer chain that never arrives
    dev.gn_step()
    got = dev.get("step")                                 # idto_hip_get repeats the solve on the variants below
    assert dev.get_option("solver_timeouts") >= 1
    assert dev.get_option("solver_nd") == 0               # stepped down for good
    assert dev.get_option("last_solver") == 1             # the two-workgroup factorisation answered
    assert np.array_equal(got, want)
    assert np.allclose(first, want, rtol=0, atol=1e-3 * np.abs(want).max())   # (two factorisations differ ~cond * eps)
    dev.set_option("debug_skip_role", -1)
    dev.gn_step()
    assert np.array_equal(dev.get("step"), want) and not dev.solver_status()[0]
    dev.close()


def test_the_iteration_kernels_own_wait_times_out_and_the_loop_steps_down(monkeypatch):
    """tr_iter_kernel's workgroups poll each other's sums (csrc/trust_region.h): one that never publishes - option
    debug_skip_role 100 + block row - is what a partner that is not resident looks like.  The others give up after their
    bound, the launch reports IDTO_HIP_SOLVER_TIMEOUT (the count's upper half: not a solver's wait), the resident loop
    is off for the context (tr_resident_ok), the solvers are left alone; TrajectoryOptimizer::Solve runs again on the loop
    that returns to the host twice an iteration and walks the iterates of an undisturbed optimizer."""
    from idto_amd.optimizer import TrajectoryOptimizer, TrajectoryOptimizerSolution, TrajectoryOptimizerStats
    from idto_amd.problem import SCALING
    cfg, model = load_config("hopper"), load_model("hopper")
    prob, sp, q_guess = make_problem(cfg, model, num_steps=20)
    sp.equality_constraints = False
    q = synthetic_trajectory(cfg, model, 20, seed=3, lower=0.01)
    dev = hip.HipPath(model, prob, sp)
    dev.set_q(q)
    dev.eval_tau()
    dev.set_option("debug_skip_role", 100 + 7)
    with pytest.raises(hip.SolverTimeout):
        dev.tr_solve(3, SCALING["double_sqrt"], True, False, 1e-1, 1e5)
    assert dev.get_option("solver_timeouts") == 1 and dev.get_option("tr_resident_ok") == 0
    assert dev.get_option("solver_nd") == 1 and dev.get_option("solver_pipe") == 1   # (the solvers keep their variants)
    assert np.array_equal(dev.get("q"), q)                                            # nothing was accepted on garbage
    with pytest.raises(RuntimeError):
        dev.tr_solve(3, SCALING["double_sqrt"], True, False, 1e-1, 1e5)               # refused from now on
    dev.close()
    # through the optimizer: the second attempt takes the stepwise loop, same iterates as an undisturbed run
    sp.max_iterations, sp.verbose = 6, False
    ref = TrajectoryOptimizer(model, prob, sp)
    rs, rt = TrajectoryOptimizerSolution(), TrajectoryOptimizerStats()
    ref.Solve(q_guess, rs, rt)
    monkeypatch.setenv("IDTO_DEBUG_SKIP_ROLE", str(100 + 7))   # (read when the context is created)
    opt = TrajectoryOptimizer(model, prob, sp)
    monkeypatch.delenv("IDTO_DEBUG_SKIP_ROLE")
    s, t = TrajectoryOptimizerSolution(), TrajectoryOptimizerStats()
    opt.Solve(q_guess, s, t)
    assert np.array_equal(s.q, rs.q) and np.array_equal(t.iteration_costs, rt.iteration_costs)
