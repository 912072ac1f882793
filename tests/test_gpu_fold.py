"""Assembly fold: fd_kernel leaves the single-record products of the Gauss-Newton assembly
(reference optimizer/trajectory_optimizer.cc:1046-1165) next to the slab and assemble_terms_kernel
only combines them.  The sums and their order are those of assemble_diag_kernel, so gradient and
Hessian bands must be bit-identical (==) with the fold on and off, for every gradients method, for
batches, and the path must fall back to the slab whenever the products do not belong to it."""
import numpy as np
import pytest

from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu

BANDS = ("gradient", "H_A", "H_B", "H_C")


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def _setup(name, N, seed=3):
    cfg, model = load_config(name), load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    sp.scaling = False
    sp.equality_constraints = False
    return cfg, model, prob, sp, synthetic_trajectory(cfg, model, N, seed=seed, lower=0.01)


@pytest.mark.parametrize("name,N", [("mini_cheetah", 40), ("allegro_hand", 20), ("hopper", 9), ("spinner", 12),
                                    ("acrobot", 8), ("mini_cheetah", 3), ("hopper", 2)])
@pytest.mark.parametrize("gradients", ["forward_differences", "central_differences", "central_differences4"])
def test_fold_is_bit_identical(name, N, gradients):
    cfg, model, prob, sp, q = _setup(name, N)
    sp.gradients_method = gradients
    out = {}
    for fold in (1, 0):
        dev = hip.HipPath(model, prob, sp)
        dev.set_option("asm_fold", fold)
        dev.set_q(q)
        dev.eval_partials()
        dev.grad_hess()
        assert dev.get_option("last_assembly") == (1 if fold else 2)
        out[fold] = {a: dev.get(a) for a in BANDS}
        dev.close()
    for a in BANDS:
        assert _same(out[1][a], out[0][a]), a
    if gradients == "forward_differences":
        g, _ = Oracle(model, prob, sp).gn_step(q)
        assert _same(out[1]["gradient"], g)


def test_fold_falls_back_when_products_are_stale():
    """a shard of the k-range, or records written through the device pointer, must be assembled from the slab"""
    cfg, model, prob, sp, q = _setup("mini_cheetah", 12)
    dev = hip.HipPath(model, prob, sp)
    dev.set_q(q)
    dev.eval_partials()
    dev.grad_hess()
    assert dev.get_option("last_assembly") == 1
    want = {a: dev.get(a) for a in BANDS}
    dev.device_ptr("slab")            # a caller that may have rewritten the records
    dev.grad_hess()
    assert dev.get_option("last_assembly") == 2
    for a in BANDS:
        assert _same(dev.get(a), want[a]), a
    dev.set_shard(0, 6)
    dev.set_q(q)
    dev.eval_partials()
    dev.set_shard(6, 12)
    dev.eval_partials()
    dev.grad_hess()
    assert dev.get_option("last_assembly") == 2
    for a in BANDS:
        assert _same(dev.get(a), want[a]), a
    dev.set_shard(0, 12)
    # new weights: the products of the old R must not be reused
    prob.R = prob.R * 3.0
    dev.set_problem(prob)
    dev.set_q(q)
    dev.eval_partials()
    dev.grad_hess()
    assert dev.get_option("last_assembly") == 1
    ref = hip.HipPath(model, prob, sp)
    ref.set_option("asm_fold", 0)
    ref.set_q(q)
    ref.eval_partials()
    ref.grad_hess()
    for a in BANDS:
        assert _same(dev.get(a), ref.get(a)), a
    ref.close()
    dev.close()


def test_fold_batch():
    cfg, model = load_config("mini_cheetah"), load_model("mini_cheetah")
    probs, qs = [], []
    for b in range(3):
        prob, sp, _ = make_problem(cfg, model, num_steps=30)
        sp.scaling = False
        sp.equality_constraints = False
        prob.R = prob.R * (1.0 + b)
        probs.append(prob)
        qs.append(synthetic_trajectory(cfg, model, 30, seed=b, lower=0.01))
    out = {}
    for fold in (1, 0):
        dev = hip.HipPath(model, probs, sp)
        dev.set_option("asm_fold", fold)
        dev.set_q_batch(np.array(qs))
        dev.gn_step()
        out[fold] = [{a: dev.get(a, b) for a in BANDS + ("step",)} for b in range(3)]
        dev.close()
    for b in range(3):
        for a in BANDS + ("step",):
            assert _same(out[1][b][a], out[0][b][a]), (b, a)


@pytest.mark.parametrize("name,N,band", [("mini_cheetah", 40, 0), ("mini_cheetah", 24, 0), ("hopper", 50, 0), ("hopper", 50, 2), ("acrobot", 40, 0),
                                         ("acrobot", 40, 1), ("spinner", 40, 0), ("spinner", 40, 1)])
def test_assembly_inside_the_solver_launch_is_bit_identical(name, N, band):
    """idto_hip_gn_step with the pipelined solver (band = 0) or the scalar band factorisation (penta_band.h): g and the
    bands are formed by workgroups of the solver's own launch (penta_pipe.h PipeAsm) - same bits as the assembly
    kernel's, and the step solved from them the same"""
    cfg, model, prob, sp, q = _setup(name, N)
    out = {}
    for inside in (1, 0):
        dev = hip.HipPath(model, prob, sp)
        dev.set_option("solver_band", band)
        dev.set_option("gn_small", 0)   # (the two-launch step is what this test is about)
        dev.set_option("asm_in_solver", inside)
        dev.set_q(q)
        for _ in range(3):   # (epoch-valued words: repeated launches)
            dev.gn_step()
        assert dev.get_option("last_solver") == (6 if band else 4)
        assert dev.get_option("last_assembly") == (4 if inside else 1)
        assert dev.solver_status() == (False, 0)
        out[inside] = {a: dev.get(a) for a in BANDS + ("step",)}
        dev.close()
    for a in BANDS + ("step",):
        assert _same(out[1][a], out[0][a]), a
    g, _ = Oracle(model, prob, sp).gn_step(q)
    assert _same(out[1]["gradient"], g)
