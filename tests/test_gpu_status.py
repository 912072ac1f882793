"""Factorisation status and solver-input bookkeeping of the device path (C-ABI):
  * a Hessian that is not numerically positive definite is REPORTED (the reference:
    PentaDiagonalFactorizationStatus::kFailure, optimizer/penta_diagonal_solver.h:181-185; the
    optimizer demands success, optimizer/trajectory_optimizer.cc:2084) and the host-side
    TrajectoryOptimizer returns SolverFlag::kFactorizationFailed
    (optimizer/trajectory_optimizer_solution.h:16-21);
  * bands written behind the API's back after an assembly are solved as a general system (the
    "block row 0 is the identity" shortcut belongs to an assembled Hessian only);
  * unsupported contact pairs are refused at creation."""
import numpy as np
import pytest
import torch

import oracle_lib as ol
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory
from test_gpu_penta import DeviceSolver
from test_oracle_penta import from_lower_dense

pytestmark = pytest.mark.gpu


def _semidefinite_problem(name="acrobot", N=12):
    """zero weight on DoF 0 everywhere and R = 0: the rows of H for that DoF are exactly zero"""
    cfg, model = load_config(name), load_model(name)
    prob, sp, q_guess = make_problem(cfg, model, num_steps=N)
    sp.scaling = False
    sp.equality_constraints = False
    sp.verbose = False
    for W in (prob.Qq, prob.Qv, prob.Qf_q, prob.Qf_v):
        W[0, :] = 0.0
        W[:, 0] = 0.0
    prob.R[:] = 0.0
    q = synthetic_trajectory(cfg, model, N, seed=3, lower=0.0)
    return model, prob, sp, q, q_guess


@pytest.mark.parametrize("two_sided", [1, 0])
@pytest.mark.parametrize("reference", [0, 1])
def test_semidefinite_hessian_is_reported(two_sided, reference):
    model, prob, sp, q, _ = _semidefinite_problem()
    dev = hip.HipPath(model, prob, sp)
    dev.set_option("two_sided", two_sided)
    dev.set_option("reference_solver", reference)
    dev.set_q(q)
    dev.gn_step()
    failed, rows = dev.solver_status()
    assert failed and rows >= 1
    with pytest.raises(hip.FactorizationFailed):
        dev.get("step")
    # host right-hand sides take the same exit
    with pytest.raises(hip.FactorizationFailed):
        dev.solve_host(np.ones((1, (dev.N + 1) * dev.nq)))
    rows = dev.solver_status()[1]
    # ... and a healthy Hessian afterwards clears the status (it belongs to the last factorisation)
    cfg = load_config("acrobot")
    prob2, _, _ = make_problem(cfg, model, num_steps=dev.N)
    dev.set_problem(prob2)
    dev.set_q(q)
    dev.gn_step()
    failed2, rows2 = dev.solver_status()
    assert not failed2 and rows2 == rows
    assert np.all(np.isfinite(dev.get("step")))
    dev.close()


def test_pivot_without_significant_digits_is_reported():
    """d_1 = (1 + 2^-52) - 1 * 1 = 2^-52: positive and finite, but it carries no digit of the
    diagonal entry it came from"""
    bs, n = 2, 12
    s = DeviceSolver(bs, n)
    Z = np.zeros((n, bs, bs))
    C = np.tile(np.eye(bs), (n, 1, 1))
    C[7] = np.array([[1.0, 1.0], [1.0, 1.0 + 2.0 ** -52]])
    s.set_bands(Z, Z, C)
    b = np.ones(n * bs)
    for two_sided in (True, False):
        s.solve(b, two_sided=two_sided)
        assert s.dev.solver_status()[0]
    C[7] = np.array([[1.0, 0.5], [0.5, 1.0]])
    s.set_bands(Z, Z, C)
    x = s.solve(b)
    assert not s.dev.solver_status()[0]
    assert np.allclose(x[14:16], np.linalg.solve(C[7], [1.0, 1.0]))


def test_optimizer_returns_factorization_failed():
    from idto_amd.optimizer import TrajectoryOptimizer, TrajectoryOptimizerSolution, TrajectoryOptimizerStats
    model, prob, sp, _, q_guess = _semidefinite_problem()
    sp.max_iterations = 5
    for method in ("trust_region", "linesearch"):
        sp.method = method
        opt = TrajectoryOptimizer(model, prob, sp)
        sol, st = TrajectoryOptimizerSolution(), TrajectoryOptimizerStats()
        assert opt.Solve(q_guess, sol, st) == "kFactorizationFailed"
        opt.close()


def test_bands_overwritten_after_assembly_are_solved_in_full():
    """grad_hess marks H as assembled (row 0 = identity, skipped by the fast solver);
    handing out the band pointer must end that assumption"""
    name, N = "hopper", 11
    cfg, model = load_config(name), load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    sp.scaling = False
    sp.equality_constraints = False
    dev = hip.HipPath(model, prob, sp)
    dev.set_q(synthetic_trajectory(cfg, model, N, seed=1, lower=0.01))
    dev.gn_step()                      # assembled Hessian: the solver starts at block row 1
    assert np.all(dev.get("step")[:model.nq] == 0.0)
    bs, n = model.nq, N + 1
    size = n * bs
    rng = np.random.default_rng(11)
    Ar = rng.uniform(-1, 1, (size, size))
    H = from_lower_dense(np.eye(size) + Ar @ Ar.T, n, bs)   # C_0 != I, B_1 != 0, A_2 != 0
    Hd = ol.penta_make_dense(*H)
    span = (n + 5) * bs * bs

    class _Ptr:
        __cuda_array_interface__ = {"shape": (3 * span,), "typestr": "<f8", "data": (dev.device_ptr("hbands"), False),
                                    "version": 2}
    view = torch.as_tensor(_Ptr(), device="cuda:0")
    host = np.zeros(3 * span)
    for k in range(3):
        host[k * span:k * span + n * bs * bs] = np.asarray(H[k]).transpose(0, 2, 1).ravel()
    view.copy_(torch.from_numpy(host))
    torch.cuda.synchronize()
    x_gt = np.linspace(-2, 3, size)
    rhs = torch.tensor((Hd @ x_gt)[None, :], dtype=torch.float64, device="cuda")
    x = torch.zeros_like(rhs)
    for nrhs_kind in ("device", "host"):
        if nrhs_kind == "device":
            dev.factor_solve(rhs.data_ptr(), 1, x.data_ptr())
            dev.sync()
            got = x.cpu().numpy()[0]
        else:
            got = dev.solve_host(rhs.cpu().numpy())[0]
        assert np.linalg.norm(got - x_gt) / np.linalg.norm(x_gt) < 100 * np.linalg.cond(Hd) * np.finfo(float).eps
    # the next assembly restores the shortcut (and x_0 = 0 for the Gauss-Newton step)
    dev.gn_step()
    assert np.all(dev.get("step")[:model.nq] == 0.0)
    dev.close()


def test_unsupported_box_box_pair_is_refused():
    """box-box distance exists only for (moving box, world-fixed axis-aligned box)"""
    import copy
    model = load_model("mini_cheetah")
    cfg = load_config("mini_cheetah")
    prob, sp, _ = make_problem(cfg, model, num_steps=4)
    box = [i for i in range(model.npairs)
           if model.geom_type[model.pair_a[i]] == 1 and model.geom_type[model.pair_b[i]] == 1]
    assert box, "mini_cheetah has a body-box vs ground-box pair"
    bad = copy.deepcopy(model)
    i = box[0]
    bad.pair_a[i], bad.pair_b[i] = model.pair_b[i], model.pair_a[i]   # (ground, body): wrong order
    with pytest.raises(hip.HipError, match="box-box"):
        hip.HipPath(bad, prob, sp)
    hip.HipPath(model, prob, sp).close()


def test_the_dense_solver_and_the_one_workgroup_step_report_it_too():
    """Round 6's two new ways to a step: `linear_solver = kDenseLdlt` (idto_hip_solve_dense_ldlt: where the reference's
    DRAKE_DEMAND(Hldlt.info() == Eigen::Success) aborts, TO.cc:2088-2093) and gn_small_kernel (acrobot's whole step in one
    workgroup, csrc/gn_small.h: the band solver's pivot test inside it)."""
    from idto_amd.optimizer import TrajectoryOptimizer, TrajectoryOptimizerSolution, TrajectoryOptimizerStats
    model, prob, sp, q, q_guess = _semidefinite_problem(N=40)
    dev = hip.HipPath(model, prob, sp)
    dev.set_q(q)
    dev.gn_step()
    assert dev.get_option("last_solver") == 7
    failed, rows = dev.solver_status()
    assert failed and rows >= 1
    with pytest.raises(hip.FactorizationFailed):
        dev.get("step")
    with pytest.raises(hip.FactorizationFailed):
        dev.solve_dense_ldlt(np.ones((dev.N + 1) * dev.nq))
    dev.close()
    sp.max_iterations, sp.linear_solver = 3, "dense_ldlt"
    opt = TrajectoryOptimizer(model, prob, sp)
    sol, st = TrajectoryOptimizerSolution(), TrajectoryOptimizerStats()
    assert opt.Solve(q_guess, sol, st) == "kFactorizationFailed"
    opt.close()


def test_mpc_replan_with_a_failed_factorisation_is_not_silent():
    """the C++ controller keeps the previous plan when a re-plan's factorisation fails and says so through
    last_flag(); the Python wrapper raises (strict, the default) instead of handing the stale plan back as if it were new."""
    from idto_amd.mpc import DeviceModelPredictiveController
    from idto_amd.optimizer import TrajectoryOptimizer, TrajectoryOptimizerSolution
    model, prob, sp, q, q_guess = _semidefinite_problem(N=20)
    sp.max_iterations = 1
    opt = TrajectoryOptimizer(model, prob, sp)
    plan = TrajectoryOptimizerSolution()   # (any plan to shift: the re-plan's factorisation is what fails)
    plan.q, plan.v, plan.tau = np.asarray(q_guess, float), np.zeros((21, model.nv)), np.zeros((20, model.nv))
    x = np.concatenate([plan.q[0], plan.v[0]])
    strict = DeviceModelPredictiveController(opt, plan, actuated=model.actuated, replan_period=0.01)
    with pytest.raises(RuntimeError, match="factorisation failed"):
        strict.update(0.01, x[:model.nq], x[model.nq:])
    assert strict.last_flag == 2
    strict.close()
    lenient = DeviceModelPredictiveController(opt, plan, actuated=model.actuated, replan_period=0.01, strict=False)
    g, qq, v, tau = lenient.update(0.01, x[:model.nq], x[model.nq:])
    assert lenient.last_flag == 2 and np.isfinite(qq).all()
    lenient.close(); opt.close()
