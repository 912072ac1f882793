"""Device-side trust-region bookkeeping (include/idto_hip.h idto_hip_tr_*) against
NumPy on the arrays the device path itself reports: scale factors (CalcScaleFactors,
optimizer/trajectory_optimizer.cc:1225-1255), the inner products CalcDoglegPoint (:2108-2202) and
CalcTrustRatio (:1979-2035) are made of, the trial point q + dq with tau / cost evaluated there,
acceptance, rejection and the speculative launch of the next iteration."""
import numpy as np
import pytest

import oracle_lib as ol
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import SCALING, load_config, make_problem, synthetic_trajectory
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu


def _setup(name, N, seed=1):
    cfg, model = load_config(name), load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    sp.equality_constraints = False
    return cfg, model, prob, sp, synthetic_trajectory(cfg, model, N, seed=seed, lower=0.01)


@pytest.mark.parametrize("name,N,method", [("mini_cheetah", 40, "double_sqrt"), ("hopper", 12, "sqrt"),
                                           ("allegro_hand", 10, None), ("acrobot", 9, "adaptive_double_sqrt")])
def test_prepare_trial_accept(name, N, method):
    cfg, model, prob, sp, q = _setup(name, N)
    dev = hip.HipPath(model, prob, sp)
    dev.set_unactuated_dofs(model.unactuated_dofs)
    dev.set_q(q)
    dev.eval_tau()
    cost0 = dev.get("cost")
    dev.gn_step()
    sm = -1 if method is None else SCALING[method]
    S = dev.tr_prepare(sm)
    g, step = dev.get("gradient"), dev.get("step")
    bands = [dev.get(k) for k in ("H_A", "H_B", "H_C")]
    Cs, Dm, Em = ol.penta_make_symmetric(*bands)
    H = ol.penta_make_dense(bands[0], bands[1], Cs, Dm, Em)
    d = np.diag(H)
    if method is None:
        D = np.ones_like(d)
    elif "double" in method:
        D = np.minimum(1.0, 1.0 / np.sqrt(np.sqrt(d)))
    else:
        D = np.minimum(1.0, 1.0 / np.sqrt(d))
    if method is not None:
        assert np.array_equal(dev.get("tr_scale"), D)   # IEEE sqrt / division / min: bit-identical
    gt, w = D * g, -step / D
    Ht = D[:, None] * H * D[None, :]
    tau = dev.get("tau")
    h = tau[:, model.unactuated_dofs].ravel() if len(model.unactuated_dofs) else np.zeros(1)
    want = [gt @ gt, gt @ Ht @ gt, w @ w, gt @ w, gt @ Ht @ w, w @ Ht @ w, q.ravel() @ q.ravel(), h @ h, 0.0]
    assert np.allclose(S, want, rtol=1e-11, atol=1e-300), (S, want)
    assert np.array_equal(dev.get("tr_w"), w)
    # a dogleg-shaped step: dq = D (a g~ + b w)
    a, b = -0.3 * (S[0] / S[1]), -0.6
    T = dev.tr_trial(a, b, method is not None)
    dq = D * (a * gt + b * w)
    assert np.array_equal(dev.get("tr_dq"), dq)
    assert np.isclose(T[0], dq @ dq, rtol=1e-12) and np.isclose(T[1], gt @ (a * gt + b * w), rtol=1e-12)
    q_trial = q + dq.reshape(q.shape)
    orc = Oracle(model, prob, sp)
    tau_t, cost_t = orc.eval_traj(q_trial)[2], orc.eval_traj(q_trial)[3]
    assert T[2] == cost_t                      # the trial point's cost, bit-identical to the oracle
    assert np.array_equal(dev.get("q"), q)     # q itself is untouched until the step is accepted
    dev.tr_accept()
    assert np.array_equal(dev.get("q"), q_trial) and np.array_equal(dev.get("tau"), tau_t) and dev.get("cost") == cost_t
    assert cost0 != cost_t
    dev.close()


def test_quaternions_are_normalised_on_request():
    cfg, model, prob, sp, q = _setup("mini_cheetah", 8)
    dev = hip.HipPath(model, prob, sp)
    dev.set_q(q)
    dev.gn_step()
    dev.tr_prepare(-1)
    dev.tr_trial(0.0, -1.0, False, normalize_quaternions=True)
    dq = dev.get("tr_dq").reshape(q.shape)
    dev.tr_accept()
    want = q + dq
    want[:, :4] /= np.linalg.norm(want[:, :4], axis=1, keepdims=True)
    got = dev.get("q")
    assert np.allclose(got, want, rtol=0, atol=1e-15) and np.abs(np.linalg.norm(got[:, :4], axis=1) - 1).max() < 1e-15
    dev.close()


@pytest.mark.parametrize("accept", [True, False])
def test_speculative_next_iteration(accept):
    """with speculation the gn_step / tr_prepare that follow an accepted step return exactly what a
    non-speculating context computes; after a rejection the old iterate's g, H are recomputed"""
    cfg, model, prob, sp, q = _setup("mini_cheetah", 40)
    sm = SCALING["double_sqrt"]
    outs = []
    for spec in (sm, -2):
        dev = hip.HipPath(model, prob, sp)
        dev.set_q(q)
        dev.gn_step()
        S0 = dev.tr_prepare(sm)
        T = dev.tr_trial(-0.1 * S0[0] / S0[1], -0.5, True, speculate_scaling_method=spec)
        if accept:
            dev.tr_accept()
        else:
            dev.tr_reject()
        dev.gn_step()
        S1 = dev.tr_prepare(sm)
        outs.append((T, S1, dev.get("gradient"), dev.get("step"), dev.get("q"), dev.get("tr_w")))
        assert dev.solver_status() == (False, 0)
        dev.close()
    for x, y in zip(outs[0], outs[1]):
        assert np.array_equal(x, y)
    if not accept:
        assert np.array_equal(outs[0][4], q) and np.array_equal(outs[0][1], S0)


def _solve(name, N, iters, stepwise, monkeypatch, scaling=True, method="double_sqrt", quat=False, q0=None):
    from idto_amd.optimizer import TrajectoryOptimizer, TrajectoryOptimizerSolution, TrajectoryOptimizerStats
    cfg, model = load_config(name), load_model(name)
    prob, sp, q_guess = make_problem(cfg, model, num_steps=N)
    sp.equality_constraints = False
    sp.max_iterations, sp.verbose = iters, False
    sp.scaling, sp.scaling_method = scaling, method
    sp.normalize_quaternions = quat
    if stepwise:
        monkeypatch.setenv("IDTO_OPT_STEPWISE", "1")
    else:
        monkeypatch.delenv("IDTO_OPT_STEPWISE", raising=False)
    opt = TrajectoryOptimizer(model, prob, sp)
    sol, st = TrajectoryOptimizerSolution(), TrajectoryOptimizerStats()
    flag = opt.Solve(q_guess if q0 is None else q0, sol, st)
    return sol, st, flag


@pytest.mark.parametrize("name,N,iters,scaling,method,quat", [
    ("mini_cheetah", 40, 12, True, "double_sqrt", False), ("mini_cheetah", 20, 8, True, "sqrt", True),
    ("hopper", 20, 15, False, "double_sqrt", False), ("allegro_hand", 30, 4, True, "double_sqrt", False),
    ("acrobot", 30, 25, True, "double_sqrt", False), ("spinner", 20, 12, True, "sqrt", False),
    # the adaptive scalings (D = min(D_prev, .), TO.cc:1241-1255): the acrobot run rejects steps, D_prev must not move then
    ("acrobot", 30, 25, True, "adaptive_double_sqrt", False), ("mini_cheetah", 40, 10, True, "adaptive_sqrt", False),
    ("hopper", 20, 15, True, "adaptive_double_sqrt", False)])
def test_resident_loop_equals_stepwise_loop(name, N, iters, scaling, method, quat, monkeypatch):
    """idto_hip_tr_solve (every iteration enqueued at once, dogleg / trust ratio / accept / radius on
    the device) walks exactly the iterates of the loop that returns to the host twice per iteration:
    same kernels for the O(num_vars) work, the same scalar expressions for the rest -> same bits"""
    a_sol, a_st, a_flag = _solve(name, N, iters, False, monkeypatch, scaling, method, quat)
    b_sol, b_st, b_flag = _solve(name, N, iters, True, monkeypatch, scaling, method, quat)
    assert a_flag == b_flag
    for series in ("iteration_costs", "trust_region_radii", "trust_ratios", "q_norms", "dq_norms", "dqH_norms",
                   "gradient_norms", "dL_dqs", "h_norms", "merits"):
        x, y = getattr(a_st, series), getattr(b_st, series)
        assert x.size == iters and np.array_equal(x, y), (series, x, y)
    assert np.array_equal(a_sol.q, b_sol.q) and np.array_equal(a_sol.v, b_sol.v) and np.array_equal(a_sol.tau, b_sol.tau)
    assert (a_st.iteration_times > 0).all() and a_st.iteration_times.sum() <= a_st.solve_time   # (device clock per iteration)
    # some steps of these runs are rejected (acrobot) - the radius shrinks and the iterate stays
    if name == "acrobot":
        assert (a_st.trust_ratios <= 0).any()


def test_resident_loop_rows_and_failure():
    """the rows idto_hip_tr_solve returns; a semidefinite Hessian is reported, not iterated on"""
    cfg, model, prob, sp, q = _setup("mini_cheetah", 24)
    dev = hip.HipPath(model, prob, sp)
    dev.set_unactuated_dofs(model.unactuated_dofs)
    dev.set_q(q)
    dev.eval_tau()
    c0 = dev.get("cost")
    rows, delta = dev.tr_solve(6, SCALING["double_sqrt"], True, False, 1e-1, 1e5)
    assert rows[0, 0] == c0 and rows[0, 1] == 1e-1 and (rows[:, 14] == 0).all()
    for k in range(1, 6):   # accepted: the next iteration starts from the trial point's cost
        assert rows[k, 0] == (rows[k - 1, 13] if rows[k - 1, 9] else rows[k - 1, 0])
    assert (np.diff(rows[:, 10]) > 0).all() and delta > 0
    assert np.array_equal(dev.get("q"), q) == (not rows[:, 9].any())
    prob.R = prob.R * 0.0
    prob.Qq = prob.Qq * 0.0
    prob.Qv = prob.Qv * 0.0
    prob.Qf_q = prob.Qf_q * 0.0
    prob.Qf_v = prob.Qf_v * 0.0
    dev.set_problem(prob)
    dev.set_q(q)
    dev.eval_tau()
    with pytest.raises(hip.FactorizationFailed):
        dev.tr_solve(3, -1, False, False, 1e-1, 1e5)
    # ... in the rows of the iterations it happened in (flag 32), none of which accepted a step
    assert (dev.last_tr_rows[:, 14].astype(int) & 32).all() and not dev.last_tr_rows[:, 9].any()
    dev.close()


@pytest.mark.parametrize("name,N,iters", [("hopper", 40, 12), ("acrobot", 40, 20), ("spinner", 40, 15), ("allegro_hand", 12, 4),
                                            ("hopper", 50, 8), ("allegro_hand", 30, 3),    # (these two: n_eq 150 / 180 > 128, blocked LDL^T)
                                            ("allegro_hand", 40, 3)])   # (the MPC horizon: 29 x 29 KKT blocks, back substitution in recursion form)
@pytest.mark.parametrize("kkt", [1, 0])
def test_resident_loop_with_equality_constraints_follows_the_host_loop(name, N, iters, kkt, monkeypatch):
    """enforced equality constraints (the example YAMLs of acrobot, spinner, hopper, allegro): the resident loop
    computes the multipliers on the device - kkt = 1: one banded solve of the KKT system (csrc/kkt.h; blocks of
    nq + nu <= 24, i.e. not allegro), kkt = 0: H^-1 [g | J^T], S = J H^-1 J^T, a single-workgroup LDL^T of S - and
    uses the merit function; the host loop forms S and uses the host's pivoted LDL^T.  Different algorithms and
    summation orders, so no bit equality, and the iteration amplifies the differences (hopper: 4e-13 after two
    iterations, 2e-7 after twelve): the tolerances of tests/test_gpu_optimizer.py's comparison with the oracle -
    costs and merits 1e-6 relative, radii exactly the same sequence of halvings / doublings, q to 1e-5."""
    monkeypatch.setenv("IDTO_CON_KKT", str(kkt))
    from idto_amd.optimizer import TrajectoryOptimizer, TrajectoryOptimizerSolution, TrajectoryOptimizerStats
    cfg, model = load_config(name), load_model(name)
    prob, sp, q_guess = make_problem(cfg, model, num_steps=N)
    sp.max_iterations, sp.verbose = iters, False
    assert sp.equality_constraints
    out = []
    for host in (False, True):
        if host:
            monkeypatch.setenv("IDTO_OPT_HOST_LOOP", "1")
        else:
            monkeypatch.delenv("IDTO_OPT_HOST_LOOP", raising=False)
        opt = TrajectoryOptimizer(model, prob, sp)
        assert opt.num_equality_constraints() > 0
        sol, st = TrajectoryOptimizerSolution(), TrajectoryOptimizerStats()
        flag = opt.Solve(q_guess, sol, st)
        out.append((flag, sol, st))
    (fa, sa, ta), (fb, sb, tb) = out
    assert fa == fb
    for series, tol in (("iteration_costs", 1e-6), ("trust_region_radii", 1e-12), ("merits", 1e-6), ("h_norms", 1e-4),
                        ("q_norms", 1e-6)):
        x, y = getattr(ta, series), getattr(tb, series)
        assert x.size == iters and np.allclose(x, y, rtol=tol, atol=1e-12), (series, x, y)
    assert np.array_equal(ta.trust_ratios > 0, tb.trust_ratios > 0)
    assert np.abs(sa.q - sb.q).max() <= 1e-5 * max(1.0, np.abs(sb.q).max())


@pytest.mark.parametrize("name,N,iters", [("acrobot", 40, 20), ("spinner", 40, 12), ("hopper", 40, 10), ("allegro_hand", 20, 3)])
def test_kkt_solution_taken_apart_inside_the_iteration_kernel(name, N, iters):
    """the banded KKT step's solution z = [-w ; lambda] is taken apart by tr_iter_kernel's workgroups (option kkt_fold,
    the default) with kkt_extract_kernel's expressions: the loop's rows, the iterate, the multipliers and J^T lambda are
    the bits of the loop that runs kkt_extract_kernel in a launch of its own"""
    cfg, model, prob, sp, q = _setup(name, N)
    out = []
    for fold in (1, 0):
        dev = hip.HipPath(model, prob, sp)
        dev.set_option("kkt_fold", fold)
        dev.set_q(q)
        dev.eval_tau()
        rows, delta = dev.tr_solve(iters, SCALING["double_sqrt"], True, False, 1e-1, 1e5, constrained_dofs=model.unactuated_dofs)
        out.append((rows.copy(), delta, dev.get("q"), dev.get("tr_w"), dev.get("tr_dq"), dev.get("tr_scale"), dev.get("con_lambda")))
        dev.close()
    a, b = out
    cols = [c for c in range(a[0].shape[1]) if c != 10]   # (column 10 is the device clock)
    assert np.array_equal(a[0][:, cols], b[0][:, cols]) and a[1] == b[1]
    assert a[0][:, 9].any() and (a[0][:, 14] == 0).all()
    for x, y in zip(a[2:], b[2:]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("kkt", [1, 0])
def test_resident_loop_flags_a_singular_constraint_system(kkt):
    """the same degree of freedom constrained twice: S = J H^-1 J^T is exactly singular; the single-workgroup
    LDL^T (kkt = 0) / the multiplier pivots of the banded KKT factorisation (kkt = 1) report it (flag 8), nothing is
    accepted afterwards and the iterate stays where it was"""
    cfg, model, prob, sp, q = _setup("hopper", 20)
    dev = hip.HipPath(model, prob, sp)
    dev.set_option("con_kkt", kkt)
    dev.set_q(q)
    dev.eval_tau()
    d = int(model.unactuated_dofs[0])
    rows, _ = dev.tr_solve(3, SCALING["double_sqrt"], True, False, 1e-1, 1e5, constrained_dofs=[d, d])
    assert (rows[:, 14].astype(int) & 8).all() and not rows[:, 9].any()
    assert np.array_equal(dev.get("q"), q)
    # and the regular set works from the same context afterwards
    dev.set_q(q)
    dev.eval_tau()
    rows, _ = dev.tr_solve(3, SCALING["double_sqrt"], True, False, 1e-1, 1e5, constrained_dofs=model.unactuated_dofs)
    assert (rows[:, 14] == 0).all() and rows[:, 9].any()
    dev.close()


@pytest.mark.parametrize("name,N", [("hopper", 40), ("acrobot", 40), ("spinner", 30), ("allegro_hand", 21), ("hopper", 9),
                                    ("acrobot", 3), ("spinner", 128), ("acrobot", 65), ("hopper", 42)])   # n_eq = 3, 128, 65, 126
def test_single_workgroup_multiplier_solve(name, N):
    """constraint_lambda_kernel: lambda = S^-1 (h - J H^-1 g) by an unpivoted LDL^T in one workgroup, against an
    extended-precision solution of the same system (S, J H^-1 g as the device formed them).  S = J H^-1 J^T is badly
    conditioned on these models, so the statement is the backward-stable one: the residual is at rounding level,
    and the error against the refined solution is no worse than that of LAPACK's pivoted LU on the same S."""
    cfg, model, prob, sp, q = _setup(name, N)
    dev = hip.HipPath(model, prob, sp)
    dev.set_option("con_kkt", 0)
    dofs = np.asarray(model.unactuated_dofs)
    dev.set_q(q)
    dev.eval_tau()
    h = dev.get("tau")[:, dofs].ravel()
    dev.tr_solve(1, SCALING["double_sqrt"], True, False, 1e-1, 1e5, constrained_dofs=dofs)
    neq = dofs.size * N
    raw = dev.get("con_S")
    S, Jy = raw[:neq * neq].reshape(neq, neq).T, raw[neq * neq:]
    lam = dev.get("con_lambda")
    r = h - Jy
    assert np.abs(S - S.T).max() <= 1e-9 * np.abs(S).max()
    res = np.abs(S @ lam - r).max() / (np.abs(S).max() * np.abs(lam).max() + np.abs(r).max())
    assert res <= 1e-12, res
    want, unc = ol.refined_solution(0.5 * (S + S.T), r)
    scale = np.abs(want).max()
    err = np.abs(lam - want).max() / scale
    err_lu = np.abs(np.linalg.solve(S, r) - want).max() / scale
    assert err <= 4 * err_lu + 16 * unc + 1e-12, (err, err_lu, unc, np.linalg.cond(S))
    dev.close()


def _kkt_reference(model, prob, sp, q, dev):
    """lambda of the KKT system [H J^T; J 0] [w; -lambda] = [g; h] in extended precision (iterative refinement with
    the residual in long double), H, g from the oracle, J, h from the device's partials (== the oracle's)"""
    import scipy.linalg as sl
    N, nq = q.shape[0] - 1, model.nq
    dofs = np.asarray(model.unactuated_dofs)
    nu = dofs.size
    g, bands = Oracle(model, prob, sp).grad_hess(q)
    H = ol.penta_make_dense(*bands)
    dev.set_q(q)
    dev.eval_partials()
    tau, dm, dt_, dp = dev.get("tau"), dev.get("dtau_dqm"), dev.get("dtau_dqt"), dev.get("dtau_dqp")
    n = (N + 1) * nq
    J = np.zeros((N * nu, n))
    for t in range(N):
        for j, d in enumerate(dofs):
            r = t * nu + j
            if t >= 2:
                J[r, (t - 1) * nq:t * nq] = dm[t][d]
            if t >= 1:
                J[r, t * nq:(t + 1) * nq] = dt_[t][d]
            J[r, (t + 1) * nq:(t + 2) * nq] = dp[t][d]
    M = np.block([[H, J.T], [J, np.zeros((N * nu, N * nu))]])
    rhs = np.concatenate([g.ravel(), tau[:N][:, dofs].ravel()])
    Ml, bl = M.astype(np.longdouble), rhs.astype(np.longdouble)
    fac = sl.lu_factor(M)
    x = sl.lu_solve(fac, rhs).astype(np.longdouble)
    for _ in range(8):
        x = x + sl.lu_solve(fac, (bl - Ml @ x).astype(np.float64)).astype(np.longdouble)
    return -x[n:].astype(np.float64)


@pytest.mark.parametrize("name,N,seed", [("acrobot", 40, 1), ("spinner", 40, 1), ("hopper", 40, 1), ("hopper", 40, 2), ("hopper", 50, 3),
                                         ("hopper", 9, 1), ("acrobot", 3, 2), ("spinner", 128, 1), ("allegro_hand", 20, 1), ("allegro_hand", 60, 2)])
def test_banded_kkt_multipliers_are_as_accurate_as_the_schur_complement_chain(name, N, seed):
    """csrc/kkt.h: the multipliers from one unpivoted banded LDL^T of the KKT system against an extended-precision
    solution of that system.  The systems are badly conditioned (1e8 .. 1e12) and BOTH device paths - this one and the
    reference's route over S = J H^-1 J^T (constraints.h) - sit at the error that conditioning allows (1e-13 ..
    2e-7 relative); the statement is that the banded solve is no worse than 4x the Schur-complement chain."""
    cfg, model = load_config(name), load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    q = synthetic_trajectory(cfg, model, N, seed=seed, lower=0.01 if name == "hopper" else 0.0)
    dofs = np.asarray(model.unactuated_dofs)
    lam = {}
    for kkt in (0, 1):
        dev = hip.HipPath(model, prob, sp)
        dev.set_option("con_kkt", kkt)
        dev.set_q(q)
        dev.eval_tau()
        rows, _ = dev.tr_solve(1, SCALING["double_sqrt"], True, False, 1e-1, 1e5, constrained_dofs=dofs)
        assert rows[0, 14] == 0
        assert dev.get_option("kkt_last_solver") in ((1, 2, 6) if kkt else (0,))   # two-workgroup / seven-workgroup / scalar band factorisation
        lam[kkt] = dev.get("con_lambda")
        if kkt:
            want = _kkt_reference(model, prob, sp, q, dev)
        dev.close()
    scale = np.abs(want).max()
    e_schur, e_kkt = np.abs(lam[0] - want).max() / scale, np.abs(lam[1] - want).max() / scale
    assert e_kkt <= 4 * e_schur + 1e-12, (e_kkt, e_schur)


@pytest.mark.parametrize("name,N,constrained", [("mini_cheetah", 20, False), ("spinner", 40, True), ("hopper", 40, True), ("acrobot", 40, False)])
def test_tau_and_partials_from_one_launch_start_the_same_loop(name, N, constrained):
    """idto_hip_eval_tau_partials: v, a, N+, tau, cost and the partials of the resident q from ONE finite-difference
    launch (the derivative modes evaluate the nominal point as well) - the loop that follows is the loop that
    idto_hip_eval_tau + its own first evaluation of the partials runs, bit for bit; tau and cost are eval_tau's"""
    cfg, model = load_config(name), load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    q = synthetic_trajectory(cfg, model, N, seed=2, lower=0.01 if name in ("hopper", "mini_cheetah") else 0.0)
    dofs = np.asarray(model.unactuated_dofs) if constrained else ()
    out = {}
    for one in (1, 0):
        dev = hip.HipPath(model, prob, sp)
        dev.set_q(q)
        if one:
            dev.eval_tau_partials()
        else:
            dev.eval_tau()
        tau, cost = dev.get("tau").copy(), np.asarray(dev.get("cost")).copy()
        rows, _ = dev.tr_solve(4, SCALING["double_sqrt"], True, False, 1e-1, 1e5, constrained_dofs=dofs)
        out[one] = (tau, cost, np.delete(rows, 10, axis=1), dev.get("q").copy())   # (column 10: the clock)
        # the flag is used up: a Gauss-Newton step of the same context evaluates the partials again
        dev.set_q(q)
        dev.eval_tau_partials()
        dev.gn_step()
        dev.gn_step()
        step = dev.get("step").copy()
        dev.set_q(q)
        dev.gn_step()
        assert np.array_equal(step, dev.get("step"))
        dev.close()
    for a, b in zip(out[1], out[0]):
        assert np.array_equal(a, b)


def test_alternating_kkt_and_schur_routes_do_not_allocate_again():
    """the banded KKT step and the Schur-complement route share ONE device copy of the constrained
    degrees of freedom; alternating between them (a resident KKT solve followed by the host loop's multipliers - the
    TRF_SINGULAR_S fallback, or SolveFromWarmStart + EvalLagrangeMultipliers) for the SAME set must not allocate the
    Schur buffers again (round 4 did, ~0.25 MB per alternation here and 2 MB at allegro's size, never freed), and the
    results of both routes stay what they were."""
    import torch
    from idto_amd import hip
    cfg, model = load_config("hopper"), load_model("hopper")
    prob, sp, q0 = make_problem(cfg, model, num_steps=40)
    dofs = np.asarray(model.unactuated_dofs)
    dev = hip.HipPath(model, prob, sp)

    def cycle():
        dev.set_q(np.asarray(q0))
        dev.eval_tau()
        rows, _ = dev.tr_solve(3, 2, True, False, 1e-1, 1e5, constrained_dofs=dofs)     # banded KKT step
        dev.set_q(np.asarray(q0))
        dev.eval_partials(); dev.grad_hess()
        S, Jy = dev.constraint_schur(dofs)                                             # the reference's route
        return rows[:, :10].copy(), S.copy(), Jy.copy()

    first = cycle()
    cycle()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(40):
        out = cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert all(np.array_equal(a, b) for a, b in zip(first, out))
    assert free0 - free1 < (1 << 20), f"device memory shrank by {(free0 - free1) / 1e6:.2f} MB over 40 alternations"
    # a different set still re-sizes the buffers (and frees the superseded ones)
    S2, _ = dev.constraint_schur(dofs[:2])
    assert S2.shape == (2 * 40, 2 * 40)
    S3, _ = dev.constraint_schur(dofs)
    assert np.array_equal(S3, first[1])
    dev.close()


@pytest.mark.parametrize("name,N,iters", [("hopper", 40, 12), ("allegro_hand", 20, 4), ("acrobot", 40, 20), ("spinner", 30, 10), ("hopper", 50, 8)])
def test_kkt_system_written_by_the_assembly(name, N, iters):
    """the banded KKT system of the next iteration is written by the gated assembly as it goes (option kkt_in_asm, the
    default; kernels.h KktSink) instead of by kkt_build_kernel in a launch of its own: every entry is a copy, so rows,
    iterate and multipliers are the bits of the loop with that launch; the runs reject steps (the system then stays)."""
    cfg, model, prob, sp, q = _setup(name, N)
    out = []
    for fused in (1, 0):
        dev = hip.HipPath(model, prob, sp)
        dev.set_option("kkt_in_asm", fused)
        dev.set_option("tr_small", 0)   # (acrobot, spinner: the multi-launch loop this option lives in)
        dev.set_q(q)
        dev.eval_tau()
        rows, delta = dev.tr_solve(iters, SCALING["double_sqrt"], True, False, 1e-1, 1e5, constrained_dofs=model.unactuated_dofs)
        out.append((np.delete(rows, 10, axis=1), delta, dev.get("q"), dev.get("tau"), dev.get("con_lambda"), dev.get("tr_w")))
        dev.close()
    a, b = out
    assert a[0][:, 9].any() and (a[0][:, 13] == 0).all()
@pytest.mark.parametrize("name,N,iters,method,batch", [("mini_cheetah", 40, 10, "double_sqrt", 1), ("hopper", 40, 15, "double_sqrt", 1),
                                                        ("mini_cheetah", 24, 8, "sqrt", 1), ("hopper", 50, 12, None, 1),
                                                        ("mini_cheetah", 40, 6, "double_sqrt", 3), ("hopper", 40, 10, "adaptive_double_sqrt", 1),
                                                        # (12 x 170 workgroups of one per CU: more than the device holds at once - the ones that
                                                        # wait only wait for workgroups dispatched before them)
                                                        ("mini_cheetah", 40, 4, "double_sqrt", 12), ("hopper", 40, 8, "double_sqrt", 12)])
def test_decision_inside_the_solvers_launch(name, N, iters, method, batch):
    """the trial point's cost and the accept / reject decision by one more workgroup of the pipelined solver's launch
    (option decide_in_solver, the default; penta_pipe.h PipeAsm::decide, kernels.h cost_body - cost_kernel's work): the
    assembly's workgroups and the chains poll its word.  Rows, iterate, tau, the step: the bits of the loop that runs
    cost_kernel as a launch of its own; hopper's runs reject steps (the chains then read g and H where they are)."""
    cfg, model, prob, sp, q = _setup(name, N)
    out = []
    for inside in (1, 0):
        if batch == 1:
            dev = hip.HipPath(model, prob, sp)
            dev.set_option("decide_in_solver", inside)
            dev.set_q(q)
            dev.eval_tau()
            rows, delta = dev.tr_solve(iters, SCALING[method] if method else -1, method is not None, False, 1e-1, 1e5)
            assert dev.get_option("last_solver") == 4
        else:
            qs = [synthetic_trajectory(cfg, model, N, seed=7 + b, lower=0.01) for b in range(batch)]
            dev = hip.HipPath(model, [prob] * batch, sp)
            dev.set_option("decide_in_solver", inside)
            dev.set_q_batch(np.stack(qs))
            dev.eval_tau()
            rows, delta = dev.tr_solve_batch(iters, SCALING[method], True, False, [1e-1] * batch, 1e5)
        rows = np.delete(rows, 10, axis=-1)   # (the device clock)
        out.append((rows, np.asarray(delta)) + tuple(dev.get(n) for n in ("q", "tau", "step", "gradient", "cost")))
        dev.close()
    a, b = out
    assert a[0][..., 9].any() and (a[0][..., 13] == 0).all()
    for x, y in zip(a, b):
        assert np.array_equal(np.asarray(x), np.asarray(y))
