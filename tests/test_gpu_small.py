"""The one-workgroup Gauss-Newton step of the small models (csrc/gn_small.h: finite differences, assembly and the band
solve of acrobot / spinner in ONE launch of ONE workgroup per problem) against the two launches it stands in for
(fd_kernel + penta_band_kernel with the assembly inside, which tests/test_gpu_parity.py holds == the oracle): every
array of the step bit for bit - v, a, N+, tau, the three partial blocks (NaN block included), g, the bands, the step -
and g, H against the oracle directly.  Reference: optimizer/trajectory_optimizer.cc:178-245, :426-563, :1021-1165,
optimizer/penta_diagonal_solver.h:124-248."""
import numpy as np
import pytest

from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu
ARRAYS = ("v", "a", "nplus", "tau", "dtau_dqm", "dtau_dqt", "dtau_dqp", "gradient", "H_A", "H_B", "H_C", "step")


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def setup(name, N, seed):
    cfg, model = load_config(name), load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    sp.scaling = sp.equality_constraints = False
    q = synthetic_trajectory(cfg, model, N, seed=seed, lower=0.0)
    if name == "spinner":
        q[:, 1] = np.linspace(1.5, 1.25, N + 1)   # the finger reaches the spinner: the contact pair is active
    return cfg, model, prob, sp, q


@pytest.mark.parametrize("name", ["acrobot", "spinner"])
@pytest.mark.parametrize("N", [16, 23, 40, 64])
def test_one_workgroup_step_equals_the_two_launches(name, N):
    cfg, model, prob, sp, q = setup(name, N, seed=N)
    dev = hip.HipPath(model, prob, sp)
    dev.set_q(q)
    dev.gn_step()
    assert dev.get_option("last_solver") == 7, "gn_small_kernel did not run"
    got = {k: dev.get(k) for k in ARRAYS}
    dev.set_option("gn_small", 0)
    dev.set_q(q)
    dev.gn_step()
    assert dev.get_option("last_solver") == 6
    for k in ARRAYS:
        assert same(got[k], dev.get(k)), k
    assert dev.solver_status() == (False, 0)
    if name == "spinner":
        assert np.abs(got["tau"]).max() > 1.0   # (the contact force is in it)
    g, bands = Oracle(model, prob, sp).grad_hess(q)
    assert np.array_equal(got["gradient"].ravel(), g)
    for key, b in zip(("H_A", "H_B", "H_C"), bands[:3]):
        assert np.array_equal(got[key], b), key
    dev.close()


@pytest.mark.parametrize("name", ["acrobot", "spinner"])
def test_one_workgroup_step_in_a_batch(name):
    """grid.y = problem: three problems with different trajectories == each of them alone"""
    N, B = 40, 3
    cfg, model, prob, sp, _ = setup(name, N, seed=0)
    qs = [setup(name, N, seed=s)[4] for s in range(B)]
    bd = hip.HipPath(model, [prob] * B, sp)
    bd.set_q_batch(qs)
    bd.gn_step()
    assert bd.get_option("last_solver") == 7
    for b in range(B):
        dev = hip.HipPath(model, prob, sp)
        dev.set_q(qs[b])
        dev.gn_step()
        for k in ("tau", "gradient", "H_C", "step"):
            assert same(bd.get(k, problem=b), dev.get(k)), (b, k)
        dev.close()
    bd.close()


def test_what_the_kernel_does_not_serve_keeps_the_two_launches():
    cfg, model, prob, sp, q = setup("acrobot", 40, seed=1)
    dev = hip.HipPath(model, prob, sp)
    dev.set_q(q)
    dev.set_option("gradients_method", 1)   # central differences
    dev.gn_step()
    assert dev.get_option("last_solver") == 6
    dev.set_option("gradients_method", 0)
    dev.gn_step()
    assert dev.get_option("last_solver") == 7
    dev.close()
    cfg, model, prob, sp, q = setup("hopper", 50, seed=1)   # a planar joint, blocks of 5
    dev = hip.HipPath(model, prob, sp)
    dev.set_q(q)
    dev.gn_step()
    assert dev.get_option("last_solver") != 7
    dev.close()


@pytest.mark.parametrize("conv", [False, True])
@pytest.mark.parametrize("name,N,iters,method", [("acrobot", 40, 25, "double_sqrt"), ("spinner", 40, 15, "sqrt"), ("acrobot", 16, 12, None),
                                                  ("spinner", 23, 10, "double_sqrt"), ("acrobot", 30, 25, "adaptive_double_sqrt")])
def test_trust_region_iteration_in_one_workgroup(name, N, iters, method, conv):
    """inside idto_hip_tr_solve the small models' launch also takes the cost of the trial point and the decision (option
    tr_small, the default): per iteration tr_iter_kernel + gn_small_kernel instead of fd_kernel, cost_kernel and the solver's
    launch.  Every row of statistics (but the device clock), the iterate, tau, v, the step and the scale factors are the
    bits of the loop with the three launches; the acrobot runs reject steps (the launch then stops behind the decision)."""
    from idto_amd.problem import SCALING
    cfg, model, prob, sp, q = setup(name, N, seed=3)
    out = []
    for small, fold in ((1, 1), (1, 0), (0, 0)):   # (fold: tr_iter_kernel's part in the same launch - the iteration is ONE launch)
        dev = hip.HipPath(model, prob, sp)
        dev.set_option("tr_small", small)
        dev.set_option("tr_fold", fold)
        if conv:
            dev.tr_set_convergence([0.0, 0.0, 0.0, 0.0, 0.0, 0.0])   # (criteria that never hold: the chunks of eight iterations, the check-only pass)
        dev.set_q(q)
        dev.eval_tau()
        rows, delta = dev.tr_solve(iters, SCALING[method] if method else -1, method is not None, False, 1e-1, 1e5)
        assert dev.get_option("last_solver") == (7 if small else 6)
        out.append((rows.copy(), delta) + tuple(dev.get(n) for n in ("q", "v", "tau", "step", "gradient", "H_C", "tr_dq", "tr_w", "tr_scale", "cost")))
        dev.close()
    b = out[-1]
    cols = [c for c in range(b[0].shape[1]) if c != 10]   # (column 10 is the device clock)
    for a in out[:-1]:
        assert np.array_equal(a[0][:, cols], b[0][:, cols]) and a[1] == b[1]
        assert a[0][:, 9].any()
        if name == "acrobot" and iters >= 25:
            assert not a[0][:, 9].all(), "no step was rejected: the early exit was not exercised"
        for x, y in zip(a[2:], b[2:]):
            assert same(np.asarray(x), np.asarray(y))


@pytest.mark.parametrize("conv", [False, True])
@pytest.mark.parametrize("name,N,iters", [("acrobot", 40, 25), ("spinner", 40, 15), ("acrobot", 23, 12), ("spinner", 30, 10)])
def test_constrained_trust_region_iteration_in_one_workgroup(name, N, iters, conv):
    """... and with the example YAMLs' enforced constraint (one unactuated degree of freedom each): the launch forms the
    banded KKT system of csrc/kkt.h in LDS and solves it instead of H p = -g.  Rows, iterate, multipliers, tau: the bits
    of the loop that runs fd_kernel, cost_kernel, the assembly, kkt_build_kernel and the band solver as launches."""
    from idto_amd.problem import SCALING
    cfg, model, prob, sp, q = setup(name, N, seed=5)
    assert len(model.unactuated_dofs) == 1
    out = []
    for small, fold in ((1, 1), (1, 0), (0, 0)):
        dev = hip.HipPath(model, prob, sp)
        dev.set_option("tr_small", small)
        dev.set_option("tr_fold", fold)
        if conv:
            dev.tr_set_convergence([0.0, 0.0, 0.0, 0.0, 0.0, 0.0])   # (criteria that never hold: the chunks of eight iterations, the check-only pass)
        dev.set_q(q)
        dev.eval_tau()
        rows, delta = dev.tr_solve(iters, SCALING["double_sqrt"], True, False, 1e-1, 1e5, constrained_dofs=model.unactuated_dofs)
        out.append((rows.copy(), delta) + tuple(dev.get(n) for n in ("q", "v", "tau", "gradient", "H_C", "tr_dq", "tr_w", "tr_scale", "cost", "con_lambda")))
        dev.close()
    b = out[-1]
    cols = [c for c in range(b[0].shape[1]) if c != 10]   # (column 10 is the device clock)
    for a in out[:-1]:
        assert np.array_equal(a[0][:, cols], b[0][:, cols]) and a[1] == b[1]
        assert a[0][:, 9].any() and (a[0][:, 14] == 0).all()
        for x, y in zip(a[2:], b[2:]):
            assert same(np.asarray(x), np.asarray(y))
