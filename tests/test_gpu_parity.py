"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on the
same seeded inputs.  Bar: BIT-EXACT (==) for N+, v, a, tau, dtau/dq, cost, gradient,
Hessian bands and the Gauss-Newton step — the kernels evaluate the same fp64
expressions in the same association order with -ffp-contract=off and deterministic
sin/cos/exp/log, so any difference is a bug.  That trivially
satisfies BASELINE.json's "gradients within 1e-9 relative"."""
import numpy as np
import pytest

from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu

CASES = [  # (config, N, seed, lower)  -- BASELINE.json configs; small N first
    ("acrobot", 8, 0, 0.0),
    ("spinner", 8, 1, 0.0),
    ("hopper", 8, 2, 0.02),
    ("mini_cheetah", 6, 3, 0.03),
    ("allegro_hand", 5, 4, 0.0),
    ("acrobot", 40, 0, 0.0),
    ("spinner", 40, 0, 0.0),
    ("hopper", 50, 0, 0.01),
    ("mini_cheetah", 40, 0, 0.01),
    ("allegro_hand", 60, 0, 0.0),
]


def setup(name, N, seed, lower):
    cfg = load_config(name)
    model = load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    sp.scaling = False
    sp.equality_constraints = False
    q = synthetic_trajectory(cfg, model, N, seed=seed, lower=lower)
    if name == "spinner":  # bring the finger into contact with the spinner
        q[:, 1] = np.linspace(1.5, 1.25, N + 1)
    return model, prob, sp, q


def same(a, b):
    """bit-exact up to the sign of zero; NaN == NaN (the dtau_dqm[0] convention)"""
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


@pytest.mark.parametrize("name,N,seed,lower", CASES)
def test_hot_path_bit_exact(name, N, seed, lower):
    model, prob, sp, q = setup(name, N, seed, lower)
    orc = Oracle(model, prob, sp)
    dev = hip.HipPath(model, prob, sp)
    dev.set_q(q)

    # a1-a5: N+, v, a, tau, cost
    dev.eval_tau()
    v, a, tau, cost = orc.eval_traj(q)
    assert same(dev.get("v"), v) and same(dev.get("a"), a)
    for t in range(N + 1):
        assert same(dev.get("nplus")[t], orc.nplus(q[t]))
    assert same(dev.get("tau"), tau), np.abs(dev.get("tau") - tau).max()
    assert dev.get("cost") == cost
    if model.npairs:
        phi = np.array([orc.signed_distances(q[t])[0].min() for t in range(1, N + 1)])
        assert phi.min() < orc.contact_threshold, "test trajectory never comes near contact"

    # a8, a9: finite-difference partials
    dev.eval_partials()
    P = orc.eval_partials(q)
    for k in ("dtau_dqp", "dtau_dqt", "dtau_dqm"):
        got = dev.get(k)
        assert same(got, P[k]), (k, np.nanmax(np.abs(got - P[k])))
    assert same(dev.get("tau"), tau)

    # a11, a12: gradient and Hessian bands
    dev.grad_hess()
    g, bands = orc.grad_hess(q)
    assert same(dev.get("gradient"), g)
    assert same(dev.get("H_A"), bands[0]) and same(dev.get("H_B"), bands[1]) and same(dev.get("H_C"), bands[2])

    # a14, a15: factor + solve H p = -g
    _, p = orc.gn_step(q)
    # (i) the bit-exact restatement of the reference's pivoted-LU block Thomas
    dev.set_option("reference_solver", 1)
    dev.factor_solve()
    assert same(dev.get("step"), p), np.abs(dev.get("step") - p).max()
    # (ii) the production solver (banded block LDL^T, no pivoting): same recursion, different
    # elimination order => agreement to round-off.  "As accurate as the reference's algorithm" is
    # measured, not bounded by cond(H)*eps: both solutions are compared with an extended-precision
    # solution of the same system (oracle_lib.refined_solution: long-double iterative refinement,
    # known to `unc` relative).  Bar: the forward error of the production solver is at most 4x the
    # forward error of the pivoted-LU block Thomas (the reference's algorithm, bit-exact above),
    # and its backward error |H p + g| at most 16x.
    dev.set_option("reference_solver", 0)
    dev.factor_solve()
    p_fast = dev.get("step")
    import oracle_lib as ol
    Hd = ol.penta_make_dense(*bands)
    p_ref, unc = ol.refined_solution(Hd, -g.ravel())
    Hp = ol.penta_multiply(*bands, p)
    Hpf = ol.penta_multiply(*bands, p_fast)
    scale = np.abs(g).max() + 1e-300
    res_lu, res_fast = np.abs(Hp + g).max() / scale, np.abs(Hpf + g).max() / scale
    pn = np.abs(p_ref).max()
    err_lu, err_fast = np.abs(p.ravel() - p_ref).max() / pn, np.abs(p_fast.ravel() - p_ref).max() / pn
    _record(name, N, res_lu=res_lu, res_fast=res_fast, err_lu=err_lu, err_fast=err_fast, ref_uncertainty=unc,
            cond=np.linalg.cond(Hd))
    assert res_fast <= 16 * res_lu + 1e-13, (res_fast, res_lu)
    assert err_fast <= 4 * err_lu + 16 * unc + 1e-12, (err_fast, err_lu, unc)

    # the fused entry point gives the same answer
    dev.set_q(q)
    dev.gn_step()
    assert same(dev.get("step"), p_fast) and same(dev.get("gradient"), g)
    dev.close()


_RECORDS = []


def _record(name, N, **kw):
    import json
    import os
    _RECORDS.append(dict(config=name, N=N, **{k: float(v) for k, v in kw.items()}))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/solver_accuracy.json", "w") as f:
        json.dump(_RECORDS, f, indent=1)


def test_device_arithmetic_is_ieee():
    """sqrt, division and idto::detmath on the device agree bit for bit with the host."""
    import oracle_lib as ol
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-40, 40, 200000), np.exp(rng.uniform(-30, 30, 100000)), rng.normal(size=100000)])
    x = x[x != 0]
    out = hip.math_probe(x)
    assert np.array_equal(out["sqrt"], np.sqrt(np.abs(x)))
    assert np.array_equal(out["recip"], 1.0 / x)
    s, c = ol.det_sincos(x)
    assert np.array_equal(out["sin"], s) and np.array_equal(out["cos"], c)
    assert np.array_equal(out["exp"], ol.det_exp(x))
    assert np.array_equal(out["log"], ol.det_log(np.abs(x)))


@pytest.mark.parametrize("name,N", [("hopper", 12), ("hopper", 9), ("hopper", 10), ("spinner", 8), ("mini_cheetah", 40),
                                    ("allegro_hand", 11), ("acrobot", 25)])
def test_multi_rhs_factor_solve(name, N):
    """H X = R for many right-hand sides: column 0 inside the factorisation kernel, the others by
    the substitution kernel walking the one- or two-sided factors (n = N + 1 >= 10: two-sided)."""
    import torch
    import oracle_lib as ol
    model, prob, sp, q = setup(name, N, 5, 0.01)
    orc = Oracle(model, prob, sp)
    dev = hip.HipPath(model, prob, sp)
    dev.set_q(q)
    dev.eval_partials()
    dev.grad_hess()
    nvars = (N + 1) * model.nq
    rng = np.random.default_rng(1)
    _, bands = orc.grad_hess(q)
    Hd = ol.penta_make_dense(*bands)
    for nrhs in (3, 70):
        rhs = torch.tensor(rng.normal(size=(nrhs, nvars)), dtype=torch.float64, device="cuda")
        x = torch.zeros_like(rhs)
        xe = ol.penta_solve(*bands, rhs.cpu().numpy())
        dev.set_option("reference_solver", 1)
        dev.factor_solve(rhs.data_ptr(), nrhs, x.data_ptr())
        dev.sync()
        assert np.array_equal(x.cpu().numpy(), xe)
        dev.set_option("reference_solver", 0)
        rn = rhs.cpu().numpy()
        res_lu = np.abs(xe @ Hd - rn).max() / np.abs(rn).max()   # H symmetric
        for two_sided in (1, 0):
            dev.set_option("two_sided", two_sided)
            x.zero_()
            dev.factor_solve(rhs.data_ptr(), nrhs, x.data_ptr())
            dev.sync()
            xf = x.cpu().numpy()
            res_fast = np.abs(xf @ Hd - rn).max() / np.abs(rn).max()
            assert res_fast <= 16 * res_lu + 1e-13, (two_sided, res_fast, res_lu)
    dev.close()


def test_update_problem_and_shard():
    name, N = "mini_cheetah", 8
    model, prob, sp, q = setup(name, N, 7, 0.02)
    dev = hip.HipPath(model, prob, sp)
    dev.set_q(q)
    dev.gn_step()
    p_full, slab_full = dev.get("step"), dev.get("slab")
    # sharded evaluation of the partials in two halves gives the same slab
    dev2 = hip.HipPath(model, prob, sp)
    dev2.set_q(q)
    dev2.set_shard(0, N // 2)
    dev2.eval_partials()
    dev2.set_shard(N // 2, N)
    dev2.eval_partials()
    s2 = dev2.get("slab")
    assert np.all((s2 == slab_full) | (np.isnan(s2) & np.isnan(slab_full)))
    dev2.grad_hess()
    dev2.factor_solve()
    assert np.array_equal(dev2.get("step"), p_full)  # same kernels, same inputs
    # UpdateNominalTrajectory changes the gradient
    prob.q_nom = prob.q_nom + 0.1
    dev.set_problem(prob)
    dev.gn_step()
    orc = Oracle(model, prob, sp)
    g, p = orc.gn_step(q)
    assert np.array_equal(dev.get("gradient"), g)
    import oracle_lib as ol
    _, bands = orc.grad_hess(q)
    p_ref, unc = ol.refined_solution(ol.penta_make_dense(*bands), -g.ravel())
    pn = np.abs(p_ref).max()
    err_lu, err_fast = np.abs(p.ravel() - p_ref).max() / pn, np.abs(dev.get("step").ravel() - p_ref).max() / pn
    assert err_fast <= 4 * err_lu + 16 * unc + 1e-12, (err_fast, err_lu, unc)
    dev.set_option("reference_solver", 1)
    dev.factor_solve()
    assert np.array_equal(dev.get("step"), p)
    dev.close()
    dev2.close()


@pytest.mark.parametrize("name,N", [("spinner", 1), ("spinner", 2), ("spinner", 3), ("hopper", 4), ("hopper", 9),
                                     ("hopper", 10), ("free_body", 5), ("pendulum", 12), ("acrobot", 3)])
def test_edge_horizons_and_padded_blocks(name, N):
    """Shortest horizons (N = 1: a single tau), the one-sided / two-sided switch of the solver
    (n = N + 1 = 10), and block sizes that are padded to a template size (pendulum nq = 1 and
    free_body nq = 7 run in K = 8 blocks): same bit-exact bar as the main cases."""
    from idto_amd.problem import ProblemDefinition, SolverParameters
    model = load_model(name)
    nq, nv = model.nq, model.nv
    rng = np.random.default_rng(N)
    if name in ("free_body", "pendulum"):
        q0 = np.array([1.0, 0, 0, 0, 0.1, 0.2, 0.3]) if name == "free_body" else np.array([0.3])
        prob = ProblemDefinition(num_steps=N, q_init=q0, v_init=np.zeros(nv), Qq=np.eye(nq), Qv=0.1 * np.eye(nv),
                                 Qf_q=10 * np.eye(nq), Qf_v=np.eye(nv), R=0.5 * np.eye(nv),
                                 q_nom=np.tile(q0, (N + 1, 1)), v_nom=np.zeros((N + 1, nv)), time_step=0.05)
        sp = SolverParameters(verbose=False, scaling=False, equality_constraints=False)
        q = np.tile(q0, (N + 1, 1)) + 0.05 * rng.normal(size=(N + 1, nq))
        q[0] = q0
    else:
        model, prob, sp, q = setup(name, N, N, 0.01)
    orc = Oracle(model, prob, sp)
    dev = hip.HipPath(model, prob, sp)
    dev.set_q(q)
    dev.gn_step()
    v, a, tau, cost = orc.eval_traj(q)
    g, bands = orc.grad_hess(q)
    _, p = orc.gn_step(q)
    assert same(dev.get("tau"), tau) and same(dev.get("v"), v)
    P = orc.eval_partials(q)
    for key in ("dtau_dqp", "dtau_dqt", "dtau_dqm"):
        assert same(dev.get(key), P[key]), key
    assert same(dev.get("gradient"), g)
    assert same(dev.get("H_A"), bands[0]) and same(dev.get("H_B"), bands[1]) and same(dev.get("H_C"), bands[2])
    p_fast = dev.get("step")
    import oracle_lib as ol
    scale = np.abs(g).max() + 1e-300
    res = lambda x: np.abs(ol.penta_multiply(*bands, x) + g).max() / scale
    assert res(p_fast) <= 16 * res(p) + 1e-13, (res(p_fast), res(p))
    dev.set_option("reference_solver", 1)
    dev.factor_solve()
    assert same(dev.get("step"), p)
    dev.eval_tau()
    assert dev.get("cost") == cost
    dev.close()


@pytest.mark.parametrize("method", ["central_differences", "central_differences4"])
@pytest.mark.parametrize("name,N,seed,lower", [("acrobot", 8, 0, 0.0), ("spinner", 10, 1, 0.0), ("hopper", 12, 2, 0.02),
                                              ("mini_cheetah", 6, 3, 0.03), ("allegro_hand", 5, 4, 0.0),
                                              ("mini_cheetah", 40, 0, 0.01)])
def test_central_difference_partials_bit_exact(name, N, seed, lower, method):
    """SolverParameters::gradients_method = kCentralDifferences / kCentralDifferences4
    (reference trajectory_optimizer.cc:565-885): the device evaluates tau at q_t[i] +- dq
    (and +- 2 dq) in the same expressions as the oracle; partials, gradient, Hessian bands
    and the step of the reference-order solver are bit-identical."""
    model, prob, sp, q = setup(name, N, seed, lower)
    sp.gradients_method = method
    orc = Oracle(model, prob, sp)
    dev = hip.HipPath(model, prob, sp)
    dev.set_q(q)
    dev.eval_partials()
    P = orc.eval_partials(q)
    for k in ("dtau_dqp", "dtau_dqt", "dtau_dqm"):
        got = dev.get(k)
        assert same(got, P[k]), (k, np.nanmax(np.abs(got - P[k])))
    _, _, tau, _ = orc.eval_traj(q)
    assert same(dev.get("tau"), tau)
    dev.grad_hess()
    g, bands = orc.grad_hess(q)
    assert same(dev.get("gradient"), g)
    assert same(dev.get("H_A"), bands[0]) and same(dev.get("H_B"), bands[1]) and same(dev.get("H_C"), bands[2])
    # and the methods agree with forward differences to truncation error
    sp.gradients_method = "forward_differences"
    dev_f = hip.HipPath(model, prob, sp)
    dev_f.set_q(q)
    dev_f.eval_partials()
    a, b = dev.get("dtau_dqt"), dev_f.get("dtau_dqt")
    assert np.abs(a - b).max() <= 2e-4 * (1.0 + np.abs(b).max())


def test_trial_cost_equals_set_q_eval_tau():
    """idto_hip_trial_cost (one call, one synchronisation) == set_q + eval_tau + get(tau) + get(cost),
    bit for bit, and leaves the context ready for eval_partials on the same q"""
    model, prob, sp, q = setup("mini_cheetah", 12, 5, 0.02)
    orc = Oracle(model, prob, sp)
    dev = hip.HipPath(model, prob, sp)
    tau, cost = dev.trial_cost(q)
    _, _, tau_ref, cost_ref = orc.eval_traj(q)
    assert same(tau, tau_ref) and cost == cost_ref
    dev.eval_partials()
    P = orc.eval_partials(q)
    assert same(dev.get("dtau_dqt"), P["dtau_dqt"])
    q2 = q + 1e-3
    tau2, cost2 = dev.trial_cost(q2)   # second call reuses the staging buffers
    _, _, tau2_ref, cost2_ref = orc.eval_traj(q2)
    assert same(tau2, tau2_ref) and cost2 == cost2_ref


def test_prefetch_semantics():
    """idto_hip_prefetch: get() returns the array as of the prefetch even though later kernels
    have been launched (and are still running); a prefetch that became stale is dropped"""
    model, prob, sp, q = setup("mini_cheetah", 20, 2, 0.02)
    dev = hip.HipPath(model, prob, sp)
    dev.set_q(q)
    dev.eval_partials(); dev.grad_hess()
    g_ref, H_ref = dev.get("gradient"), dev.get("hbands")
    # enqueue the copies, then more work on the main stream, then read
    dev.grad_hess()
    dev.prefetch("gradient"); dev.prefetch("hbands")
    dev.factor_solve()
    assert np.array_equal(dev.get("gradient"), g_ref) and np.array_equal(dev.get("hbands"), H_ref)
    p = dev.get("step")
    assert np.all(np.isfinite(p))
    # stale prefetch: the gradient is recomputed for another q before it is read
    dev.prefetch("gradient")
    q2 = q.copy(); q2[1:] += 1e-3
    dev.set_q(q2); dev.eval_partials(); dev.grad_hess()
    g2 = dev.get("gradient")
    assert not np.array_equal(g2, g_ref)
    dev.sync()
    assert np.array_equal(g2, dev.get("gradient"))
    dev.close()
