"""idto::detmath (include/idto/detmath.h) against libm: the deterministic sin/cos,
exp and log the hot path uses must be accurate to ~1 ulp on the domains the
physics visits (joint angles, -phi/sigma exponents)."""
import numpy as np

from oracle_lib import det_exp, det_log, det_sincos


def _ulp_err(got, ref):
    return np.abs(got - ref) / np.spacing(np.abs(ref))


def test_sincos_accuracy():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-50, 50, 200000), rng.uniform(-1e4, 1e4, 50000), np.linspace(-7, 7, 20001)])
    s, c = det_sincos(x)
    # absolute error near zeros of sin/cos is bounded by the reduction (~1e-16 * |x|/pi)
    tol = 2.3e-16 * np.maximum(1.0, np.abs(x) / 100.0)
    assert np.all(np.abs(s - np.sin(x)) <= tol)
    assert np.all(np.abs(c - np.cos(x)) <= tol)
    # exact identities the kinematics relies on
    s0, c0 = det_sincos(np.array([0.0]))
    assert s0[0] == 0.0 and c0[0] == 1.0


def test_exp_accuracy():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-60, 60, 200000), np.linspace(-700, 700, 20001), [0.0, 1e-10, -1e-10, 37.0]])
    y = det_exp(x)
    assert np.max(_ulp_err(y, np.exp(x))) <= 1.5
    assert det_exp(np.array([0.0]))[0] == 1.0


def test_log_accuracy():
    rng = np.random.default_rng(2)
    x = np.concatenate([np.exp(rng.uniform(-60, 60, 200000)), 1.0 + rng.uniform(-1e-3, 1e-3, 20000), [1.0, 2.0]])
    y = det_log(x)
    ref = np.log(x)
    assert np.all(np.abs(y - ref) <= 1.5 * np.spacing(np.maximum(np.abs(ref), 1e-300)) + 1e-19)
    assert det_log(np.array([1.0]))[0] == 0.0


def test_softplus_branch_limit():
    # reference optimizer/trajectory_optimizer.cc:351-359: x = 37 is the first integer with exp(x)+1 == exp(x)
    e = det_exp(np.array([37.0, 36.0]))
    assert e[0] + 1.0 == e[0]
    assert e[1] + 1.0 != e[1]
