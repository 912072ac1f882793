"""The host-side TrajectoryOptimizer (libidto_opt.so -> libidto_hip.so) against the reference's
own Python tests (python_bindings/test/trajectory_optimizer_test.py, warm_start_test.py), the
property tests of optimizer/test/trajectory_optimizer_test.cc ("TO_test.cc") and the CPU
oracle's Solve on identical problems.

The outer loop reuses one device factorisation per state and the production solver is the
banded LDL^T, so iterates agree with the oracle to round-off amplified by cond(H), not bit for
bit: tolerances are stated per test."""
import numpy as np
import pytest

from idto_amd.model import load_model
from idto_amd.optimizer import (TrajectoryOptimizer, TrajectoryOptimizerSolution, TrajectoryOptimizerStats)
from idto_amd.problem import ProblemDefinition, SolverParameters, load_config, make_problem
from oracle_lib import Oracle
from test_oracle_trajopt import mk as mk_oracle
from test_oracle_trajopt import pendulum, spinner_python_test_problem

pytestmark = pytest.mark.gpu
EPS = np.finfo(float).eps
SQRT_EPS = np.sqrt(EPS)


def solve(opt, q_guess):
    sol, st = TrajectoryOptimizerSolution(), TrajectoryOptimizerStats()
    flag = opt.Solve(q_guess, sol, st)
    return sol, st, flag


def test_spinner_end_to_end_golden():
    """expected_qN = [0.287, 1.497, 1.995] +- 1e-3 "from CPP version"
    (python_bindings/test/trajectory_optimizer_test.py:17-91)"""
    model, prob, sp, q_guess = spinner_python_test_problem()
    opt = TrajectoryOptimizer(model, prob, sp)
    assert opt.time_step() == 0.05 and opt.num_steps() == 40
    sol, st, flag = solve(opt, q_guess)
    assert sol.q.shape == (41, 3) and sol.v.shape == (41, 3) and sol.tau.shape == (40, 3)
    assert np.linalg.norm(sol.q[-1] - np.array([0.287, 1.497, 1.995])) < 1e-3
    assert flag == "kMaxIterationsReached" and st.iteration_costs.size == 200
    assert st.solve_time > 0


@pytest.mark.parametrize("name,iters,method", [("acrobot", 10, None), ("spinner", 10, None), ("hopper", 10, None),
                                                ("mini_cheetah", 5, None), ("allegro_hand", 3, None),
                                                ("spinner", 6, "central_differences"),
                                                ("hopper", 5, "central_differences4")])
def test_solve_tracks_the_oracle(name, iters, method):
    """Every example config (reference examples/*/*.yaml; scaling and equality constraints ON as
    in the YAMLs): the device-backed Solve follows the CPU oracle's iterates.  Tolerance: cost and
    trust-region radius sequences equal to 1e-6 relative; the reference's own Python test accepts
    1e-3 on q for CPP vs Python."""
    cfg = load_config(name)
    model = load_model(name)
    prob, sp, q_guess = make_problem(cfg, model)
    sp.max_iterations, sp.verbose, sp.num_threads = iters, False, 1
    if method:  # the airhockey example's choice (reference examples/airhockey/airhockey.yaml)
        sp.gradients_method = method
    ref = Oracle(model, prob, sp).solve(q_guess)
    opt = TrajectoryOptimizer(model, prob, sp)
    assert opt.num_equality_constraints() == Oracle(model, prob, sp).num_eq
    sol, st, flag = solve(opt, q_guess)
    assert st.iteration_costs.size == iters and flag == "kMaxIterationsReached"
    rc = ref["stats"]
    assert np.allclose(st.iteration_costs, rc.iteration_costs, rtol=1e-6), (st.iteration_costs, rc.iteration_costs)
    assert np.allclose(st.trust_region_radii, rc.trust_region_radii, rtol=1e-12)
    assert np.allclose(st.h_norms, rc.h_norms, rtol=1e-5, atol=1e-9)
    assert np.abs(sol.q - ref["q"]).max() <= 1e-5 * max(1.0, np.abs(ref["q"]).max())
    assert st.iteration_costs[-1] <= st.iteration_costs[0]


def test_eval_matches_oracle():
    """gradient bit-exact (device == oracle); scaling, multipliers and merit to round-off"""
    cfg = load_config("hopper")
    model = load_model("hopper")
    prob, sp, q_guess = make_problem(cfg, model, num_steps=12)
    sp.verbose = False
    from idto_amd.problem import synthetic_trajectory
    q = synthetic_trajectory(cfg, model, 12, seed=2, lower=0.02)
    opt, orc = TrajectoryOptimizer(model, prob, sp), Oracle(model, prob, sp)
    got, want = opt.eval(q), orc.eval_all(q)
    g_ref, _ = orc.grad_hess(q)
    assert got["cost"] == orc.eval_traj(q)[3]
    assert np.array_equal(got["gradient"], g_ref)
    assert np.array_equal(got["scale_factors"], want["D"])
    assert np.array_equal(got["scaled_gradient"], want["g_scaled"])
    lam, lam_ref = got["lagrange_multipliers"], want["lam"]
    assert lam.size == orc.num_eq > 0
    assert np.abs(lam - lam_ref).max() <= 1e-6 * np.abs(lam_ref).max()
    assert abs(got["merit"] - want["merit"]) <= 1e-8 * abs(want["merit"])
    assert np.abs(got["merit_gradient"] - want["merit_grad"]).max() <= 1e-6 * np.abs(want["merit_grad"]).max()


def test_dogleg_point():  # TO_test.cc:285-362
    N, dt = 2, 5e-2
    model = pendulum()
    o, prob, sp = mk_oracle(model, N, dt, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 1.0, 0.0, 0.0, scaling=False)
    opt = TrajectoryOptimizer(model, prob, sp)
    q = np.array([[0.0], [1.5], [1.5]])
    tol = EPS / dt
    dq_s, _, act = opt.dogleg(q, 1e-3)
    assert act and abs(np.linalg.norm(dq_s) - 1e-3) < tol
    dq_l, _, act = opt.dogleg(q, 1e3)
    assert not act and np.linalg.norm(dq_l) > np.linalg.norm(dq_s)
    dq_m, _, act = opt.dogleg(q, 1.0)
    assert act and abs(np.linalg.norm(dq_m) - 1.0) < tol
    assert np.linalg.norm(dq_l) > np.linalg.norm(dq_m) > np.linalg.norm(dq_s)


def test_trust_ratio_is_one_for_linear_system():  # TO_test.cc:369-429
    N, dt = 5, 5e-2
    model = pendulum(False)
    o, prob, sp = mk_oracle(model, N, dt, 0.1, 0.0, 1.0, 2.0, 3.0, 4.0, 5.0, np.pi, -0.3)
    opt = TrajectoryOptimizer(model, prob, sp)
    q = np.array([0.1 + 0.01 * t for t in range(N + 1)]).reshape(-1, 1)
    dq, _, active = opt.dogleg(q, 1e3)   # inside the trust region: the full Gauss-Newton step, unscaled
    assert not active
    assert abs(opt.trust_ratio(q, dq) - 1.0) < SQRT_EPS


@pytest.mark.parametrize("target", [np.pi, -1.2])
def test_pendulum_swingup_and_update_nominal(target):  # TO_test.cc:434-490, 1754-1827
    N, dt = 20, 5e-2
    model = pendulum()
    _, prob, sp = mk_oracle(model, N, dt, 0.1, 0.0, 1.0, 0.1, 1000, 1, 0.01, np.pi, 0.0, max_iterations=20,
                            check_convergence=True, rel_cost_reduction=1e-5)
    opt = TrajectoryOptimizer(model, prob, sp)
    if target != np.pi:
        opt.UpdateNominalTrajectory(np.full((N + 1, 1), target), np.zeros((N + 1, 1)))
    sol, st, flag = solve(opt, np.full((N + 1, 1), 0.1))
    assert flag == "kSuccess"
    assert abs(sol.q[N, 0] - target) < 1e-3


def test_warm_start_equivalence():
    """10 iterations in one Solve == 10 x SolveFromWarmStart(max_iterations=1)
    (python_bindings/test/warm_start_test.py:165-182; same process, same device arithmetic =>
    identical to round-off)"""
    model, prob, sp, q_guess = spinner_python_test_problem()
    sp.max_iterations = 10
    _, full, _ = solve(TrajectoryOptimizer(model, prob, sp), q_guess)
    sp1 = SolverParameters(**{**sp.__dict__, "max_iterations": 1})
    o1 = TrajectoryOptimizer(model, prob, sp1)
    ws = o1.CreateWarmStart(q_guess)
    costs, radii = [], []
    for _ in range(10):
        sol, st = TrajectoryOptimizerSolution(), TrajectoryOptimizerStats()
        o1.SolveFromWarmStart(ws, sol, st)
        costs.append(st.iteration_costs[0])
        radii.append(st.trust_region_radii[0])
    assert np.allclose(costs, full.iteration_costs, rtol=0, atol=1e-8)
    assert np.allclose(radii, full.trust_region_radii, rtol=0, atol=1e-8)
    assert np.array_equal(ws.get_q(), sol.q)


@pytest.mark.parametrize("constrained", [False, True])
def test_adaptive_scaling_memory_belongs_to_the_state(constrained):
    """The adaptive scalings take min(previous D, new D) (TO.cc:1241-1255) and the previous D lives in the
    state: a second Solve on the same optimizer starts from ones again (identical statistics, and the
    oracle's), while warm starts carry it over (ten one-iteration warm starts == one ten-iteration Solve)."""
    cfg, model = load_config("spinner"), load_model("spinner")
    prob, sp, q_guess = make_problem(cfg, model)
    sp.max_iterations, sp.verbose, sp.scaling, sp.scaling_method = 8, False, True, "adaptive_double_sqrt"
    sp.equality_constraints = constrained
    ref = Oracle(model, prob, sp).solve(q_guess)["stats"]
    opt = TrajectoryOptimizer(model, prob, sp)
    sol1, st1, _ = solve(opt, q_guess)
    sol2, st2, _ = solve(opt, q_guess)
    assert np.array_equal(st1.iteration_costs, st2.iteration_costs) and np.array_equal(sol1.q, sol2.q)
    assert np.array_equal(st1.trust_region_radii, st2.trust_region_radii)
    assert np.allclose(st1.iteration_costs, ref.iteration_costs, rtol=1e-6)
    sp1 = SolverParameters(**{**sp.__dict__, "max_iterations": 1})
    o1 = TrajectoryOptimizer(model, prob, sp1)
    ws = o1.CreateWarmStart(q_guess)
    costs = []
    for _ in range(8):
        sol, st = TrajectoryOptimizerSolution(), TrajectoryOptimizerStats()
        o1.SolveFromWarmStart(ws, sol, st)
        costs.append(st.iteration_costs[0])
    assert np.allclose(costs, st1.iteration_costs, rtol=0, atol=1e-8)


def test_reset_initial_conditions():  # python_bindings/test/warm_start_test.py:119-139
    model, prob, sp, q_guess = spinner_python_test_problem()
    sp.max_iterations = 2
    opt = TrajectoryOptimizer(model, prob, sp)
    q0, v0 = np.array([0.35, 1.45, 0.05]), np.array([0.1, -0.1, 0.2])
    opt.ResetInitialConditions(q0, v0)
    ws = opt.CreateWarmStart(np.tile(q0, (41, 1)))
    sol, st = TrajectoryOptimizerSolution(), TrajectoryOptimizerStats()
    opt.SolveFromWarmStart(ws, sol, st)
    assert np.array_equal(sol.q[0], q0) and np.array_equal(sol.v[0], v0)


def test_unsupported_options_fail_loudly():
    """Every SolverParameters field the build does not honour is rejected in the constructor (reference
    optimizer/solver_parameters.h:64-167; include/idto/optimizer/trajectory_optimizer.h lists them)."""
    for field, value, match in (("gradients_method", "autodiff", "finite-difference"),
                                ("exact_hessian", True, "exact_hessian"),
                                ("save_contour_data", True, "plotting"), ("save_lineplot_data", True, "plotting"),
                                ("linesearch_plot_every_iteration", True, "plotting")):
        model, prob, sp, _ = spinner_python_test_problem()
        setattr(sp, field, value)
        with pytest.raises(RuntimeError, match=match):
            TrajectoryOptimizer(model, prob, sp)


@pytest.mark.parametrize("name,method,eq,scaling", [("spinner", "trust_region", False, True),
                                                    ("hopper", "trust_region", True, True),
                                                    ("mini_cheetah", "trust_region", True, False),
                                                    ("acrobot", "linesearch", False, False)])
def test_dense_ldlt_linear_solver_tracks_the_oracle(name, method, eq, scaling):
    """SolverParameters::linear_solver = kDenseLdlt (reference SolveLinearSystemInPlace, TO.cc:2088-2093:
    H.MakeDense().ldlt().solve(b)) is honoured: the dogleg's Newton step (:2140) and the linesearch direction
    (:2302) come from a dense LDL^T of MakeDense() on the device (idto_hip_solve_dense_ldlt).  Checked against
    the oracle run with the same setting (oracle/traj_opt.h:711-716, pivoted dense LDL^T on the CPU)."""
    cfg = load_config(name)
    model = load_model(name)
    prob, sp, q_guess = make_problem(cfg, model, num_steps=20)
    sp.max_iterations, sp.verbose, sp.num_threads = 5, False, 1
    sp.method, sp.equality_constraints, sp.scaling = method, eq, scaling
    sp.linear_solver = "dense_ldlt"
    ref = Oracle(model, prob, sp).solve(q_guess)
    opt = TrajectoryOptimizer(model, prob, sp)
    calls0 = dense_solve_count()
    sol, st, flag = solve(opt, q_guess)
    # (one per accepted iterate: a rejected step keeps H and the cached Newton step)
    assert dense_solve_count() - calls0 >= 2, "the dense LDL^T was not the solver that ran"
    rc = ref["stats"]
    assert st.iteration_costs.size == len(rc.iteration_costs)
    assert np.allclose(st.iteration_costs, rc.iteration_costs, rtol=1e-6), (st.iteration_costs, rc.iteration_costs)
    assert np.abs(sol.q - ref["q"]).max() <= 1e-5 * max(1.0, np.abs(ref["q"]).max())
    # ... and the default solver's iterates are the same to the conditioning of H (the two branches solve one system)
    sp.linear_solver = "pentadiagonal_lu"
    calls1 = dense_solve_count()
    sol2, st2, _ = solve(TrajectoryOptimizer(model, prob, sp), q_guess)
    assert dense_solve_count() == calls1
    assert np.allclose(st.iteration_costs, st2.iteration_costs, rtol=1e-6)


def dense_solve_count():
    from idto_amd import hip
    hip.lib().idto_hip_dense_solve_count.restype = __import__("ctypes").c_long
    return hip.lib().idto_hip_dense_solve_count()


@pytest.mark.parametrize("name,N", [("mini_cheetah", 40), ("hopper", 50), ("acrobot", 40), ("allegro_hand", 12)])
def test_dense_ldlt_solve_accuracy(name, N):
    """idto_hip_solve_dense_ldlt alone: H x = b against the extended-precision solution, next to the
    block-Thomas production solver on the same Hessian (what debug_compare_against_dense prints)."""
    import oracle_lib as ol
    from idto_amd import hip
    from idto_amd.problem import synthetic_trajectory
    cfg, model = load_config(name), load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    sp.scaling = sp.equality_constraints = False
    q = synthetic_trajectory(cfg, model, N, seed=1, lower=0.02)
    dev = hip.HipPath(model, prob, sp, device=0)
    dev.set_q(q)
    dev.gn_step()
    g = dev.get("gradient").ravel()
    bands = Oracle(model, prob, sp).grad_hess(q)[1]
    x_ref, unc = ol.refined_solution(ol.penta_make_dense(*bands), g)
    xd, xt = dev.solve_dense_ldlt(g), dev.solve_host(g).ravel()
    dev.set_option("reference_solver", 1)   # the reference's default branch: block Thomas with pivoted LU
    dev.gn_step()
    x_lu = -dev.get("step").ravel()
    scale = np.abs(x_ref).max()
    err_d, err_t, err_lu = (np.abs(x - x_ref).max() / scale for x in (xd, xt, x_lu))
    # the criterion of tests/test_gpu_solver_accuracy.py: no further from the exact solution than 4x the reference's own
    # default solver (+ what is known about the exact solution)
    assert err_d <= 4 * err_lu + 16 * unc + 1e-12, (err_d, err_lu, err_t, unc)
    # ... and the two device solvers agree to that accuracy ("Sparse vs. Dense error", TO.cc:2142-2150)
    assert np.abs(xd - xt).max() / scale <= 8 * err_lu + 32 * unc + 1e-12
    dev.close()


def test_debug_switches_print_what_the_reference_prints(capfd):
    """debug_compare_against_dense (TO.cc:2142-2150) and print_debug_data (:2499-2507) on the hopper: the relative
    distance of the banded solve from the dense one per dogleg point, and Eigen's rcond-style condition estimates."""
    import re
    cfg, model = load_config("hopper"), load_model("hopper")
    prob, sp, q_guess = make_problem(cfg, model, num_steps=20)
    sp.max_iterations, sp.verbose = 3, False
    sp.debug_compare_against_dense = sp.print_debug_data = True
    opt = TrajectoryOptimizer(model, prob, sp)
    sol, st, flag = solve(opt, q_guess)
    out = capfd.readouterr().out
    errs = [float(x) for x in re.findall(r"Sparse vs. Dense error: (\S+)", out)]
    assert len(errs) == 3 and all(0 <= e < 1e-6 for e in errs), out
    conds = [float(x) for x in re.findall(r"condition_number = (\S+)", out)]
    scaled = [float(x) for x in re.findall(r"condition_number_scaled = (\S+)", out)]
    assert len(conds) == 3 and len(scaled) == 3
    # the estimate is a lower bound of the 1-norm condition number, within a small factor of it in practice
    orc = Oracle(model, prob, sp)
    import oracle_lib as ol
    Hd = ol.penta_make_dense(*orc.grad_hess(q_guess)[1])
    true = np.linalg.cond(Hd, 1)
    assert true / 10 <= conds[0] <= true * 1.001, (conds[0], true)
    assert all(sc < c for sc, c in zip(scaled, conds)), "scaling must improve the conditioning (TO.cc:1212-1223)"
    # switches off: same iterates through the resident loop
    sp.debug_compare_against_dense = sp.print_debug_data = False
    sol2, st2, _ = solve(TrajectoryOptimizer(model, prob, sp), q_guess)
    assert np.allclose(st.iteration_costs, st2.iteration_costs, rtol=1e-6)


@pytest.mark.parametrize("name,ls,eq", [("acrobot", "armijo", False), ("acrobot", "backtracking", False),
                                         ("spinner", "armijo", False), ("hopper", "backtracking", True),
                                         ("mini_cheetah", "armijo", False)])
def test_linesearch_method_tracks_the_oracle(name, ls, eq):
    """SolverMethod::kLinesearch (reference SolveWithLinesearch, TO.cc:2244-2407, Armijo :1931-1977
    and backtracking :1852-1929): every linesearch iteration is one device trial point; the
    iterates follow the CPU oracle's (same alphas, same number of linesearch iterations)."""
    cfg = load_config(name)
    model = load_model(name)
    prob, sp, q_guess = make_problem(cfg, model, num_steps=20)
    sp.max_iterations, sp.verbose, sp.num_threads = 6, False, 1
    sp.method, sp.linesearch_method = "linesearch", ls
    sp.scaling, sp.equality_constraints = False, eq
    ref = Oracle(model, prob, sp).solve(q_guess)
    opt = TrajectoryOptimizer(model, prob, sp)
    sol, st, flag = solve(opt, q_guess)
    rc = ref["stats"]
    from idto_amd.optimizer import SOLVER_FLAGS
    assert flag == SOLVER_FLAGS[ref["flag"]]
    assert st.iteration_costs.size == len(rc.iteration_costs)
    assert np.array_equal(st.linesearch_iterations, np.asarray(rc.linesearch_iterations))
    assert np.allclose(st.linesearch_alphas, rc.linesearch_alphas, rtol=1e-12)
    # (six iterations of a line search amplify last-bit differences of the linear solves: the hopper with its
    # constraints, cond(H) ~ 1e10, goes from 5e-11 after one iteration to 1e-6 after five; the solves themselves are
    # checked against an extended-precision solution in tests/test_gpu_parity.py and tests/test_gpu_penta.py)
    assert np.allclose(st.iteration_costs, rc.iteration_costs, rtol=1e-5)
    assert np.all(np.isnan(st.trust_region_radii))
    assert np.abs(sol.q - ref["q"]).max() <= 1e-5 * max(1.0, np.abs(ref["q"]).max())
    assert st.iteration_costs[-1] < st.iteration_costs[0]


@pytest.mark.parametrize("name,N", [("spinner", 12), ("hopper", 20), ("hopper", 50), ("allegro_hand", 12),
                                    ("mini_cheetah", 40)])
def test_constraint_step_on_the_device(name, N):
    """idto_hip_constraint_solve (S = J H^-1 J^T factorised on the device, blocked LDL^T) against
    dense linear algebra on the oracle's H and J: lambda, H^-1 (g + J^T lambda), J^T lambda."""
    import oracle_lib as ol
    from idto_amd import hip
    from idto_amd.problem import synthetic_trajectory
    cfg, model = load_config(name), load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    sp.scaling, sp.equality_constraints = False, True
    q = synthetic_trajectory(cfg, model, N, seed=3, lower=0.02)
    orc = Oracle(model, prob, sp)
    dofs = [j for j in range(model.nv) if not model.actuated[j]] or [0, 1]
    g, bands = orc.grad_hess(q)
    H = ol.penta_make_dense(*bands)
    P = orc.eval_partials(q)
    nq, nv, neq = model.nq, model.nv, len(dofs) * N
    J = np.zeros((neq, (N + 1) * nq))
    for t in range(N):
        for i, dof in enumerate(dofs):
            r = t * len(dofs) + i
            J[r, (t + 1) * nq:(t + 2) * nq] = P["dtau_dqp"][t][dof]      # blocks are (nv, nq)
            if t > 0:
                J[r, t * nq:(t + 1) * nq] = P["dtau_dqt"][t][dof]
            if t > 1:
                J[r, (t - 1) * nq:t * nq] = P["dtau_dqm"][t][dof]
    _, _, tau, _ = orc.eval_traj(q)
    h = np.asarray(tau).reshape(N, nv)[:, dofs].ravel()
    Y = np.linalg.solve(H, np.column_stack([g, J.T]))
    S = J @ Y[:, 1:]
    lam_ref = np.linalg.solve(S, h - J @ Y[:, 0])
    step_ref = Y[:, 0] + Y[:, 1:] @ lam_ref
    dev = hip.HipPath(model, prob, sp)
    dev.set_q(q); dev.eval_partials(); dev.grad_hess()
    ok, lam, step, jtl = dev.constraint_solve(dofs, h)
    assert ok
    # both sides solve with H (cond up to 1e10 for the hopper) and S: round-off ~ cond * eps
    tol = max(1e-9, 50 * np.finfo(float).eps * max(np.linalg.cond(S), np.linalg.cond(H)))
    assert np.abs(lam - lam_ref).max() <= tol * np.abs(lam_ref).max()
    assert np.abs(jtl - J.T @ lam_ref).max() <= tol * np.abs(J.T @ lam_ref).max()
    assert np.abs(step - step_ref).max() <= tol * np.abs(step_ref).max()
    # the two-call path (host factorisation) gives the same multipliers
    S_dev, Jy_dev = dev.constraint_schur(dofs)
    assert np.abs(S_dev - S).max() <= 1e-8 * np.abs(S).max()
    lam2 = np.linalg.solve(S_dev, h - Jy_dev)
    assert np.abs(lam2 - lam).max() <= tol * np.abs(lam).max()
    dev.close()


def test_constraint_solve_reports_a_singular_schur_complement():
    """redundant constraints (the same dof listed twice: two identical rows of J) make S = J H^-1 J^T
    exactly singular: the device factorisation (no pivoting) must say so instead of returning
    garbage, and the two-call path still serves the caller's pivoted solve"""
    from idto_amd import hip
    from idto_amd.problem import synthetic_trajectory
    cfg, model = load_config("hopper"), load_model("hopper")
    N = 10
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    q = synthetic_trajectory(cfg, model, N, seed=1, lower=0.02)
    dev = hip.HipPath(model, prob, sp)
    dev.set_q(q); dev.eval_partials(); dev.grad_hess()
    dofs = [0, 1, 1]
    ok, lam, step, jtl = dev.constraint_solve(dofs, np.zeros(len(dofs) * N))
    assert not ok
    S, Jy = dev.constraint_schur(dofs)   # S is formed again from Y (the factorisation overwrote it)
    assert np.all(np.isfinite(S)) and np.abs(S - S.T).max() <= 1e-9 * np.abs(S).max()
    assert np.allclose(S[1::3, :], S[2::3, :], rtol=1e-12, atol=0)   # duplicated rows
    lam = np.linalg.lstsq(S, -Jy, rcond=None)[0]
    step, jtl = dev.constraint_step(lam)
    assert np.all(np.isfinite(step)) and np.all(np.isfinite(jtl))
    dev.close()


def test_equality_constraints_and_scaling_invariants():  # TO_test.cc:1637-1751, on the device path
    """hopper without ground: the multipliers do not depend on the scaling and equal the dense
    formula (J H^-1 J^T)^-1 (h - J H^-1 g); the merit function is the same; the scaled merit
    gradient is D times the unscaled one; the trust ratio of the Newton step is the same and > 0.6"""
    import oracle_lib as ol
    from idto_amd.problem import ProblemDefinition, SolverParameters
    model = load_model("hopper_no_ground")
    N, dt = 5, 1e-2
    q_init = np.array([0.0, 0.6, 0.3, -0.5, 0.2])
    v_init = np.array([1.0, -0.2, 0.1, -0.3, 0.4])
    prob = ProblemDefinition(num_steps=N, q_init=q_init, v_init=v_init, Qq=0.1 * np.eye(5), Qv=0.2 * np.eye(5),
                             Qf_q=0.3 * np.eye(5), Qf_v=0.4 * np.eye(5), R=0.01 * np.eye(5),
                             q_nom=np.tile([0.5, 0.5, 0.3, -0.4, 0.1], (N + 1, 1)),
                             v_nom=np.tile([0.01, 0.0, 0.2, 0.1, -0.1], (N + 1, 1)), time_step=dt)
    sp_u = SolverParameters(verbose=False, scaling=False, equality_constraints=True)
    sp_s = SolverParameters(verbose=False, scaling=True, equality_constraints=True)
    opt_u, opt_s = TrajectoryOptimizer(model, prob, sp_u), TrajectoryOptimizer(model, prob, sp_s)
    q = np.array([q_init + dt * t * v_init for t in range(N + 1)])
    e, es = opt_u.eval(q), opt_s.eval(q)
    D = es["scale_factors"]
    assert np.all(e["scale_factors"] == 1.0) or np.allclose(e["scaled_gradient"], e["gradient"])
    # dense formula with the oracle's H, J (bit-identical to the device's g, H, dtau/dq)
    orc = Oracle(model, prob, sp_u)
    ref = orc.eval_all(q)
    g, bands = orc.grad_hess(q)
    Hinv = np.linalg.inv(ol.penta_make_dense(*bands))
    J, h = ref["J"], ref["h"]
    lam_dense = np.linalg.solve(J @ Hinv @ J.T, h - J @ Hinv @ g)
    lscale = max(1.0, np.abs(lam_dense).max())
    assert e["lagrange_multipliers"].size == 3 * N
    assert np.abs(lam_dense - e["lagrange_multipliers"]).max() <= 1e-8 * lscale
    assert np.abs(lam_dense - es["lagrange_multipliers"]).max() <= 1e-8 * lscale
    assert abs(e["merit"] - es["merit"]) <= 1e-9 * max(1.0, abs(e["merit"]))
    assert np.abs(D * e["merit_gradient"] - es["merit_gradient"]).max() <= 1.5e-8 * max(1.0, np.abs(es["merit_gradient"]).max())
    dq = -Hinv @ e["merit_gradient"]
    rho, rho_s = opt_u.trust_ratio(q, dq), opt_s.trust_ratio(q, dq)
    assert rho > 0.6
    assert abs(rho - rho_s) <= 1e-7


@pytest.mark.parametrize("name,tols,constrained", [
    ("spinner", dict(rel_cost_reduction=1e-4), False),          # stops on the cost criterion
    ("spinner", dict(rel_state_change=2e-3), False),            # ... on the state criterion
    ("acrobot", dict(rel_gradient_along_dq=1e-3), False),      # ... on the gradient criterion (after rejected steps)
    ("hopper", dict(rel_cost_reduction=3e-3), True),            # with enforced equality constraints (merit gradient)
    ("mini_cheetah", dict(rel_cost_reduction=0.019), False),    # third iteration
    ("spinner", dict(), False),                                 # zero tolerances are never satisfied: runs to max_iterations
])
def test_convergence_criteria_inside_the_device_loop(name, tols, constrained, monkeypatch):
    """check_convergence = true with the device-resident loop (VerifyConvergenceCriteria, TO.cc:2654-2689, evaluated by
    the iteration kernel that follows an accepted step): the solve stops at the iteration the CPU oracle stops at, with
    the oracle's flag and reason bitmask, and the host loop (IDTO_OPT_HOST_LOOP=1) agrees"""
    cfg, model = load_config(name), load_model(name)
    prob, sp, q_guess = make_problem(cfg, model)
    sp.max_iterations, sp.verbose, sp.num_threads = 40 if name != "mini_cheetah" else 12, False, 1
    sp.equality_constraints = constrained
    sp.check_convergence = True
    for k in ("rel_cost_reduction", "abs_cost_reduction", "rel_gradient_along_dq", "abs_gradient_along_dq",
              "rel_state_change", "abs_state_change"):
        setattr(sp, k, tols.get(k, 0.0))
    ref = Oracle(model, prob, sp).solve(q_guess)
    n_ref = ref["stats"].iteration_costs.size
    out = {}
    for host in (False, True):
        if host:
            monkeypatch.setenv("IDTO_OPT_HOST_LOOP", "1")
        else:
            monkeypatch.delenv("IDTO_OPT_HOST_LOOP", raising=False)
        opt = TrajectoryOptimizer(model, prob, sp)
        sol, st, flag = solve(opt, q_guess)
        out[host] = (sol, st, flag, opt.last_convergence_reason)
        opt.close()
    flags = {0: "kSuccess", 3: "kMaxIterationsReached"}
    for host in (False, True):
        sol, st, flag, reason = out[host]
        assert st.iteration_costs.size == n_ref, (host, st.iteration_costs.size, n_ref)
        assert flag == flags.get(ref["flag"], flag) and reason == ref["reason"], (host, flag, reason, ref["flag"], ref["reason"])
        assert np.allclose(st.iteration_costs, ref["stats"].iteration_costs, rtol=1e-6)
        assert np.abs(sol.q - ref["q"]).max() <= 1e-5 * max(1.0, np.abs(ref["q"]).max())
    if tols:
        assert n_ref < sp.max_iterations and out[False][3] != 0
    else:
        assert n_ref == sp.max_iterations and out[False][2] == "kMaxIterationsReached"
    assert np.array_equal(out[False][1].trust_region_radii, out[True][1].trust_region_radii) or \
        np.allclose(out[False][1].trust_region_radii, out[True][1].trust_region_radii, rtol=1e-12)
