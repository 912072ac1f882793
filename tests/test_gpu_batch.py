"""Batch of problems on one device (idto_hip_create_batch / idto_hip_gn_step_batch; the
production call pattern of the reference's MPC examples is one Gauss-Newton iteration per
control tick per warm-started problem, examples/mpc_controller.cc:43-85; BASELINE config 5 is a
batch of 8 allegro problems): every problem of the batch must come out bit-identical (==) to the
same problem run alone in a single-problem context - and therefore to the oracle."""
import numpy as np
import pytest

from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu

ARRAYS = ("v", "a", "tau", "nplus", "dtau_dqm", "dtau_dqt", "dtau_dqp", "gradient", "H_A", "H_B", "H_C", "step")


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def _problems(name, N, B):
    cfg, model = load_config(name), load_model(name)
    probs, qs = [], []
    sp = None
    for b in range(B):
        prob, sp, _ = make_problem(cfg, model, num_steps=N)
        sp.scaling = False
        sp.equality_constraints = False
        rng = np.random.default_rng(100 + b)
        prob.q_nom = prob.q_nom + 0.01 * b                      # different nominal trajectories ...
        prob.v_init = prob.v_init + 0.05 * rng.normal(size=model.nv)   # ... initial velocities ...
        prob.Qq = prob.Qq * (1.0 + 0.1 * b)                     # ... and weights
        probs.append(prob)
        qs.append(synthetic_trajectory(cfg, model, N, seed=b, lower=0.01))
    return model, probs, sp, np.array(qs)


@pytest.mark.parametrize("name,N,B", [("mini_cheetah", 40, 5), ("allegro_hand", 60, 3), ("hopper", 9, 4),
                                      ("spinner", 12, 7), ("acrobot", 8, 2)])
@pytest.mark.parametrize("reference_solver", [0, 1])
def test_batch_equals_single_problem_contexts(name, N, B, reference_solver):
    model, probs, sp, qs = _problems(name, N, B)
    batch = hip.HipPath(model, probs, sp)
    assert batch.batch == B
    batch.set_option("reference_solver", reference_solver)
    batch.set_q_batch(qs)
    batch.eval_tau()
    costs = [batch.get("cost", b) for b in range(B)]
    batch.gn_step()
    assert batch.solver_status_batch() == [False] * B
    for b in range(B):
        one = hip.HipPath(model, probs[b], sp)
        one.set_option("reference_solver", reference_solver)
        one.set_q(qs[b])
        one.eval_tau()
        assert one.get("cost") == costs[b]
        one.gn_step()
        for arr in ARRAYS:
            assert _same(batch.get(arr, b), one.get(arr)), (b, arr)
        one.close()
    # problem 0 through the plain accessors
    assert _same(batch.get("step"), batch.get("step", 0))
    # and against the oracle for the last problem (reference-order solver: bit-exact step)
    orc = Oracle(model, probs[B - 1], sp)
    g, p = orc.gn_step(qs[B - 1])
    assert _same(batch.get("gradient", B - 1), g)
    if reference_solver:
        assert _same(batch.get("step", B - 1), p)
    batch.close()


def test_batch_problem_update_and_status():
    """set_problem_batch replaces one problem's data; a semidefinite problem in the batch is reported
    for that problem only"""
    name, N, B = "acrobot", 12, 3
    model, probs, sp, qs = _problems(name, N, B)
    batch = hip.HipPath(model, probs, sp)
    batch.set_q_batch(qs)
    batch.gn_step()
    before = [batch.get("step", b) for b in range(B)]
    import copy
    bad = copy.deepcopy(probs[1])
    for W in (bad.Qq, bad.Qv, bad.Qf_q, bad.Qf_v):
        W[0, :] = 0.0
        W[:, 0] = 0.0
    bad.R[:] = 0.0
    batch.set_problem_batch(1, bad)
    batch.gn_step()
    assert batch.solver_status_batch() == [False, True, False]
    with pytest.raises(hip.FactorizationFailed):
        batch.get("step", 1)
    assert _same(batch.get("step", 0), before[0]) and _same(batch.get("step", 2), before[2])
    batch.set_problem_batch(1, probs[1])
    batch.gn_step()
    assert batch.solver_status_batch() == [False] * B
    assert _same(batch.get("step", 1), before[1])
    batch.close()


@pytest.mark.parametrize("name,N,B,iters,method", [("mini_cheetah", 40, 5, 8, "double_sqrt"), ("allegro_hand", 60, 8, 4, "double_sqrt"),
                                                   ("acrobot", 30, 6, 20, "sqrt"), ("hopper", 20, 3, 10, "adaptive_double_sqrt"),
                                                   ("spinner", 20, 4, 10, None)])
def test_trust_region_loop_of_a_batch_equals_the_single_problem_loops(name, N, B, iters, method):
    """idto_hip_tr_solve_batch: B problems (different nominal trajectories, weights, initial guesses, radii) advance
    through the whole trust-region loop in ONE launch set per iteration from one host thread; each accepts / rejects on
    its own.  Rows, final radius and final iterate of every problem == those of idto_hip_tr_solve on the same problem
    in a context of its own (BASELINE config 5 / examples/mpc_controller.cc:43-85: several warm-started problems per tick)."""
    from idto_amd.problem import SCALING
    model, probs, sp, qs = _problems(name, N, B)
    sm = SCALING[method] if method else -1
    d0 = np.array([1e-1 * (1 + 0.5 * b) for b in range(B)])
    # (convergence criteria on for two of the cases: a problem that converges idles while the others go on)
    conv = [1e-2, 0.0, 0.0, 0.0, 1e-3, 0.0] if name in ("mini_cheetah", "spinner") else None
    want = []
    for b in range(B):
        dev = hip.HipPath(model, probs[b], sp)
        # (a batch keeps blocks of 3 on the block kernels; one problem alone takes the scalar band
        # factorisation from 16 block rows on: same solver on both sides, or the comparison is between roundings)
        if name == "spinner":
            dev.set_option("solver_band", 0)
        dev.tr_set_convergence(conv)
        dev.set_q(qs[b])
        dev.eval_tau()
        rows, delta = dev.tr_solve(iters, sm, method is not None, False, d0[b], 1e5)
        want.append((rows, delta, dev.get("q"), dev.get("v"), dev.get("tau")))
        dev.close()
    bd = hip.HipPath(model, probs, sp)
    if name == "spinner":
        bd.set_option("solver_band", 0)
    bd.tr_set_convergence(conv)
    bd.set_q_batch(qs)
    bd.eval_tau()
    rows, delta = bd.tr_solve_batch(iters, sm, method is not None, False, d0, 1e5)
    if conv:
        assert any(want[b][0][:, 16].any() for b in range(B)), "no problem of the batch met a criterion: the case tests nothing"
    accepted = 0
    for b in range(B):
        r0, d, q, v, tau = want[b]
        cols = [c for c in range(17) if c != 10]   # (column 10 is the device clock)
        assert np.array_equal(rows[b][:, cols], r0[:, cols]), (b, rows[b][:, :3], r0[:, :3])
        assert delta[b] == d
        assert _same(bd.get("q", problem=b), q) and _same(bd.get("v", problem=b), v) and _same(bd.get("tau", problem=b), tau)
        accepted += int(r0[:, 9].sum())
    assert accepted > 0
    # the context is usable afterwards: a Gauss-Newton step of the batch from the final iterates == the single contexts'
    bd.gn_step()
    for b in range(B):
        dev = hip.HipPath(model, probs[b], sp)
        if name == "spinner":
            dev.set_option("solver_band", 0)
        dev.set_q(want[b][2])
        dev.gn_step()
        assert _same(bd.get("step", problem=b), dev.get("step")), b
        dev.close()
    bd.close()


@pytest.mark.parametrize("name,N,B,iters,method,kkt", [("allegro_hand", 60, 8, 3, "double_sqrt", 1), ("hopper", 20, 3, 8, "double_sqrt", 1),
                                                       ("spinner", 20, 4, 8, None, 1), ("hopper", 20, 3, 8, "double_sqrt", 0),
                                                       ("allegro_hand", 12, 2, 3, "double_sqrt", 0)])
def test_constrained_trust_region_loop_of_a_batch_equals_the_single_problem_loops(name, N, B, iters, method, kkt, monkeypatch):
    """idto_hip_tr_solve_batch_constrained: the batch's trust-region loop with the equality constraints ENFORCED on the
    unactuated degrees of freedom (h = tau[unactuated] = 0 and its multipliers, reference TO.cc:1267-1396) - what BASELINE
    config 5 iterates (examples/allegro_hand/allegro_hand.yaml:95).  Rows, final radius and final iterate of every problem ==
    those of idto_hip_tr_solve with the same constraints on the same problem in a context of its own; the multiplier chain
    did run (h, column 8 of the rows, is reported and falls).  kkt = 1: the banded KKT step, one launch set per iteration
    for the whole batch; kkt = 0 (IDTO_CON_KKT=0): the Schur-complement route, a child context per problem."""
    from idto_amd.problem import SCALING
    monkeypatch.setenv("IDTO_CON_KKT", str(kkt))
    model, probs, sp, qs = _problems(name, N, B)
    dofs = list(model.unactuated_dofs)
    assert len(dofs) > 0
    sm = SCALING[method] if method else -1
    d0 = np.array([1e-1 * (1 + 0.5 * b) for b in range(B)])
    want = []
    for b in range(B):
        dev = hip.HipPath(model, probs[b], sp)
        if name == "spinner":   # (the batch keeps blocks of 3 / 4 on the block kernels: the same solver on both sides)
            dev.set_option("solver_band", 0)
        dev.set_q(qs[b])
        dev.eval_tau()
        rows, delta = dev.tr_solve(iters, sm, method is not None, False, d0[b], 1e5, constrained_dofs=dofs)
        want.append((rows, delta, dev.get("q")))
        dev.close()
    bd = hip.HipPath(model, probs, sp)
    if name == "spinner":
        bd.set_option("solver_band", 0)
    bd.set_q_batch(qs)
    for rep in range(2):   # (the second call reuses the per-problem contexts)
        if rep:
            bd.set_q_batch(qs)
        rows, delta = bd.tr_solve_batch_constrained(iters, sm, method is not None, False, d0, 1e5, dofs)
        accepted = 0
        for b in range(B):
            r0, d, q = want[b]
            cols = [c for c in range(17) if c != 10]   # (column 10 is the device clock)
            assert np.array_equal(rows[b][:, cols], r0[:, cols]), (rep, b, rows[b][:, :3], r0[:, :3])
            assert delta[b] == d
            assert _same(bd.get("q", problem=b), q)
            assert np.all(np.isfinite(r0[:, 8])) and r0[0, 8] > 0.0   # |h| of the constraints is reported
            accepted += int(r0[:, 9].sum())
        assert accepted > 0
    # the batch context goes on from the final iterates
    bd.gn_step()
    for b in range(B):
        dev = hip.HipPath(model, probs[b], sp)
        if name == "spinner":
            dev.set_option("solver_band", 0)
        dev.set_q(want[b][2])
        dev.gn_step()
        assert _same(bd.get("step", problem=b), dev.get("step")), b
        dev.close()
    bd.close()
