"""The straight-line inverse-dynamics evaluation of csrc/id_fast.h against the generic id_eval<MAXC> it specialises.

BuildModel (idto_hip.hip) recognises the tree shapes that are instantiated - acrobot, hopper, mini_cheetah, allegro_hand
(+ ball), spinner (its third body hangs off the world again and its pair touches two chain bodies: shape 5) - and
fd_kernel<MAXC, SHAPE> then evaluates with compile-time joint types, host-gathered records, contact pairs inside the
forward recursion, inputs formed by the consuming lane.  Same operations in the same order: every output of the
finite-difference kernel and of the assembly must have the same bits either way (the parity tests of
test_gpu_parity.py hold the default - the fast path where there is one - against the oracle; this file keeps the
generic path, which serves every other model, under the same bar)."""
import numpy as np
import pytest

from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu

SHAPES = {"acrobot": 1, "hopper": 2, "mini_cheetah": 3, "allegro_hand": 4, "spinner": 5}
ARRAYS = ("v", "a", "tau", "dtau_dqp", "dtau_dqt", "dtau_dqm", "gradient", "H_A", "H_B", "H_C", "step")


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def outputs(model, prob, sp, q, fast, gradients_method=0):
    dev = hip.HipPath(model, prob, sp)
    dev.set_option("fd_fast", fast)
    dev.set_option("gradients_method", gradients_method)
    dev.set_q(q)
    shape = dev.get_option("fast_shape")
    dev.gn_step()
    out = {k: dev.get(k) for k in ARRAYS}
    dev.eval_tau()
    out["tau_only"], out["cost"] = dev.get("tau"), np.array(dev.get("cost"))
    dev.close()
    return shape, out


@pytest.mark.parametrize("name,N,lower", [("acrobot", 40, 0.0), ("spinner", 40, 0.0), ("hopper", 50, 0.01), ("mini_cheetah", 40, 0.01),
                                          ("allegro_hand", 60, 0.0), ("mini_cheetah", 3, 0.02), ("allegro_hand", 2, 0.0)])
@pytest.mark.parametrize("seed", [0, 5])
def test_fast_shape_is_recognised_and_gives_the_generic_evaluation_bits(name, N, lower, seed):
    cfg, model = load_config(name), load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    sp.scaling = sp.equality_constraints = False
    q = synthetic_trajectory(cfg, model, N, seed=seed, lower=lower)
    if name == "spinner":
        q[:, 1] = np.linspace(1.5, 1.25, N + 1)
    shape, fast = outputs(model, prob, sp, q, 1)
    assert shape == SHAPES[name]
    _, generic = outputs(model, prob, sp, q, 0)
    for k in fast:
        assert same(fast[k], generic[k]), k
    if seed == 0:   # ... and the generic path against the oracle (the default path is test_gpu_parity.py's)
        orc = Oracle(model, prob, sp)
        P = orc.eval_partials(q)
        for k in ("dtau_dqp", "dtau_dqt", "dtau_dqm"):
            assert same(generic[k], P[k]), k
        assert same(generic["tau"], orc.eval_traj(q)[2])


@pytest.mark.parametrize("name,N", [("mini_cheetah", 12), ("hopper", 10), ("allegro_hand", 6)])
@pytest.mark.parametrize("method", [1, 2])
def test_central_differences_on_the_fast_shapes(name, N, method):
    """gradients_method central / central4 (reference TO.cc:565-885): the fast shapes evaluate with inputs staged in LDS"""
    cfg, model = load_config(name), load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    sp.scaling = sp.equality_constraints = False
    q = synthetic_trajectory(cfg, model, N, seed=2, lower=0.01)
    _, fast = outputs(model, prob, sp, q, 1, method)
    _, generic = outputs(model, prob, sp, q, 0, method)
    for k in fast:
        assert same(fast[k], generic[k]), k


def test_a_pair_order_the_fast_walk_cannot_keep_falls_back_to_the_generic_evaluation():
    """The fast evaluation walks a path's contact pairs slot by slot; the sum of the wrenches on the common body must keep
    the list's order (id_fast.h).  A cheetah whose body-ground pair is listed BETWEEN two pairs that touch the body through
    different feet of one path cannot be walked that way - here: all feet assigned to path 0's list is not possible, so the
    model is altered the other way: the pair list is reversed, which puts (foot, ground) pairs before (body, foot) pairs of
    a path and the body-ground pair first; BuildModel must either keep a valid walk or fall back, and the results must be
    those of the generic evaluation."""
    cfg, model = load_config("mini_cheetah"), load_model("mini_cheetah")
    import copy
    m2 = copy.deepcopy(model)
    m2.pair_a, m2.pair_b, m2.pair_path = m2.pair_a[::-1].copy(), m2.pair_b[::-1].copy(), m2.pair_path[::-1].copy()
    N = 6
    prob, sp, _ = make_problem(cfg, m2, num_steps=N)
    sp.scaling = sp.equality_constraints = False
    q = synthetic_trajectory(cfg, m2, N, seed=1, lower=0.02)
    shape, fast = outputs(m2, prob, sp, q, 1)
    _, generic = outputs(m2, prob, sp, q, 0)
    for k in fast:
        assert same(fast[k], generic[k]), (shape, k)
    orc = Oracle(m2, prob, sp)
    assert same(generic["tau"], orc.eval_traj(q)[2])


def test_spinner_pair_between_two_chain_bodies():
    """shape 5: the finger tip against the spinner - both bodies of the pair are slots of the one path; the pair is
    evaluated at the later slot and its reaction taken out of the earlier slot's wrench.  On a trajectory that keeps the
    two in contact (the contact force is there: tau changes with the stiffness) fast == generic == oracle."""
    name, N = "spinner", 40
    cfg, model = load_config(name), load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    sp.scaling = sp.equality_constraints = False
    q = synthetic_trajectory(cfg, model, N, seed=3, lower=0.0)
    q[:, 1] = np.linspace(1.5, 1.25, N + 1)
    shape, fast = outputs(model, prob, sp, q, 1)
    _, generic = outputs(model, prob, sp, q, 0)
    assert shape == 5
    for k in ARRAYS + ("tau_only", "cost"):
        assert same(fast[k], generic[k]), k
    g, bands = Oracle(model, prob, sp).grad_hess(q)
    assert same(fast["gradient"], g)
    for key, band in zip(("H_A", "H_B", "H_C"), bands[:3]):
        assert same(fast[key], band), key
    import copy
    sp2 = copy.deepcopy(sp)
    sp2.contact_stiffness = 2.0 * sp.contact_stiffness
    _, stiffer = outputs(model, prob, sp2, q, 1)
    assert not same(stiffer["tau_only"], fast["tau_only"])   # the pair is active somewhere along the trajectory
