"""idto_hip_gn_step as ONE persistent launch (csrc/fused.h: the finite-difference, assembly and
solver workgroups are roles of one grid, ordered by two device-memory counters) against the same
iteration as three dependent launches: every array of the iteration must be bit-identical, over
repeated iterations on changing trajectories (the counters are monotonic across launches)."""
import numpy as np
import pytest

from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _block_kernels_only(monkeypatch):
    """the three-launch path these tests compare with bit by bit is the block factorisation the fused launch embeds: the scalar band factorisation (penta_band.h) stays out"""
    monkeypatch.setenv("IDTO_SOLVER_BAND", "0")

ARRAYS = ("v", "a", "tau", "nplus", "dtau_dqm", "dtau_dqt", "dtau_dqp", "gradient", "H_A", "H_B", "H_C", "step")


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


@pytest.mark.parametrize("name,N", [("mini_cheetah", 40), ("allegro_hand", 60), ("hopper", 50), ("spinner", 40),
                                    ("acrobot", 40), ("hopper", 6), ("mini_cheetah", 3), ("acrobot", 2)])
def test_fused_iteration_equals_three_launches(name, N):
    cfg, model = load_config(name), load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    sp.scaling = False
    sp.equality_constraints = False
    fused, plain = hip.HipPath(model, prob, sp), hip.HipPath(model, prob, sp)
    plain.set_option("fused", 0)
    for d in (fused, plain):   # (the nested-dissection solver is a launch of its own: tests/test_gpu_nd.py)
        d.set_option("solver_nd", 0)
    for it in range(4):
        q = synthetic_trajectory(cfg, model, N, seed=it, lower=0.01)
        for d in (fused, plain):
            d.set_q(q)
            d.gn_step()
            if it == 1:
                d.gn_step()   # back-to-back launches without a host synchronisation in between
        for arr in ARRAYS:
            assert _same(fused.get(arr), plain.get(arr)), (it, arr)
        assert fused.solver_status() == (False, 0)
    # timing slot 3 is the fused launch, slots 0-2 stay empty on this path
    fused.timing_enable(1)
    fused.timing_reset()
    fused.gn_step()
    ms, n = fused.timing_get(3)
    assert n == 1 and ms > 0 and fused.timing_get(0)[1] == 0
    fused.timing_enable(False)
    # the oracle agrees with the fused path bit for bit (gradient) on the last trajectory
    g, _ = Oracle(model, prob, sp).gn_step(q)
    assert _same(fused.get("gradient"), g)
    fused.close()
    plain.close()


def test_fused_path_falls_back_when_not_eligible():
    """dense weights, a k-range shard or the reference-order solver take the three-launch path;
    results stay correct"""
    name, N = "hopper", 12
    cfg, model = load_config(name), load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    sp.scaling = False
    sp.equality_constraints = False
    q = synthetic_trajectory(cfg, model, N, seed=2, lower=0.01)
    orc = Oracle(model, prob, sp)
    g, p = orc.gn_step(q)
    dev = hip.HipPath(model, prob, sp)
    dev.set_q(q)
    dev.set_option("reference_solver", 1)
    dev.gn_step()
    assert _same(dev.get("step"), p) and _same(dev.get("gradient"), g)
    dev.set_option("reference_solver", 0)
    prob.Qq = prob.Qq + 1e-3 * np.ones_like(prob.Qq)   # dense weight matrix
    dev.set_problem(prob)
    dev.gn_step()
    g2, _ = Oracle(model, prob, sp).gn_step(q)
    assert _same(dev.get("gradient"), g2)
    dev.close()
