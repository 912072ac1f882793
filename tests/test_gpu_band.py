"""The scalar band factorisation of the small models' systems (csrc/penta_band.h, `last_solver` 6: blocks of 2 .. 5 read
as a symmetric band matrix of half width 3 K - 1, two wavefronts of one workgroup from the two ends) against (i) the
reference's linear-algebra test case (optimizer/test/penta_diagonal_solver_test.cc:188-257), (ii) the bit-exact
restatement of the reference's pivoted-LU solver and a refined solution of the oracle's Hessian, (iii) single-problem
contexts bit by bit when it runs as a batch; and the reports of a Hessian that is not positive definite, wherever in
the matrix the bad pivot sits (first chain, middle rows, mirrored chain)."""
import copy

import numpy as np
import pytest

import oracle_lib as ol
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory
from oracle_lib import Oracle
from test_gpu_penta import DeviceSolver
from test_oracle_penta import from_lower_dense

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bs", [2, 3, 5])
@pytest.mark.parametrize("n", [24, 41, 64])
def test_reference_penta_diagonal_case(bs, n):
    """penta_diagonal_solver_test.cc:188-257: an SPD block penta-diagonal system with a known solution"""
    size = n * bs
    rng = np.random.default_rng(6 + n + bs)
    Ar = rng.uniform(-1, 1, (size, size))
    H = from_lower_dense(np.eye(size) + Ar @ Ar.T, n, bs)
    Hd = ol.penta_make_dense(*H)
    s = DeviceSolver(bs, n)
    s.dev.set_option("solver_band", 2)
    s.set_bands(H[0], H[1], H[2])
    x_gt = np.linspace(-3, 12.4, size)
    tol = 50 * np.linalg.cond(Hd) * np.finfo(float).eps
    for scale in (1.0, -0.25):   # (a second launch on the same context)
        x = s.solve(scale * (Hd @ x_gt))
        assert s.dev.get_option("last_solver") == 6
        assert s.dev.solver_status() == (False, 0)
        assert np.linalg.norm(x - scale * x_gt) / np.linalg.norm(x_gt) < tol
    # many right-hand sides: the block factorisation's factors serve the substitution kernel
    X = s.solve(np.stack([Hd @ x_gt, 2 * (Hd @ x_gt)]))
    assert s.dev.get_option("last_solver") == 1
    assert np.linalg.norm(X[1] - 2 * x_gt) / np.linalg.norm(x_gt) < 2 * tol


@pytest.mark.parametrize("bs,n", [(2, 41), (3, 40), (5, 30)])
@pytest.mark.parametrize("where", ["first chain", "middle", "mirrored chain", "last row"])
def test_indefinite_matrix_is_reported(bs, n, where):
    """a negative pivot anywhere: the status of penta_diagonal_solver.h:181-185 (kFailure)"""
    size = n * bs
    rng = np.random.default_rng(3)
    Ar = rng.uniform(-1, 1, (size, size))
    Hd = ol.penta_make_dense(*from_lower_dense(np.eye(size) + Ar @ Ar.T, n, bs))
    row = {"first chain": size // 5, "middle": size // 2, "mirrored chain": (4 * size) // 5, "last row": size - 1}[where]
    Hd[row, row] = -Hd[row, row]
    H = from_lower_dense(Hd, n, bs)
    s = DeviceSolver(bs, n)
    s.dev.set_option("solver_band", 2)
    s.set_bands(H[0], H[1], H[2])
    s.solve(np.ones(size))
    assert s.dev.get_option("last_solver") == 6
    failed, rows = s.dev.solver_status()
    assert failed and rows >= 1
    # ... and the next factorisation of a healthy matrix clears it
    Hd[row, row] = -Hd[row, row]
    H = from_lower_dense(Hd, n, bs)
    s.set_bands(H[0], H[1], H[2])
    x = s.solve(Hd @ np.ones(size))
    assert not s.dev.solver_status()[0]   # (the count of failed rows is cumulative)
    assert np.allclose(x, 1.0, rtol=0, atol=1e-6)


CASES = [("acrobot", 24, 0.0, 1), ("acrobot", 31, 0.0, 1), ("acrobot", 200, 0.0, 1), ("spinner", 63, 0.0, 1), ("spinner", 126, 0.0, 1),
         ("hopper", 50, 0.01, 2), ("hopper", 100, 0.01, 2)]


@pytest.mark.parametrize("name,N,lower,band", CASES)
def test_gauss_newton_step_is_as_accurate_as_the_pivoted_lu(name, N, lower, band):
    """horizons other than the examples' (tests/test_gpu_solver_accuracy.py has those, four seeds each): the split
    into chains, the padding in front of the mirrored chain and the join move with the size"""
    cfg, model = load_config(name), load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    sp.scaling = sp.equality_constraints = False
    q = synthetic_trajectory(cfg, model, N, seed=1, lower=lower)
    if name == "spinner":
        q[:, 1] = np.linspace(1.5, 1.25, N + 1)
    g, bands = Oracle(model, prob, sp).grad_hess(q)
    p_ref, unc = ol.refined_solution(ol.penta_make_dense(*bands), -g.ravel())
    pn = np.abs(p_ref).max()
    dev = hip.HipPath(model, prob, sp)
    dev.set_option("solver_band", band)
    dev.set_option("gn_small", 0)   # (this file tests penta_band_kernel as a launch of its own; the one-workgroup step: tests/test_gpu_small.py)
    dev.set_q(q)
    dev.gn_step()
    assert dev.get_option("last_solver") == 6
    p = dev.get("step")
    dev.set_option("reference_solver", 1)
    dev.gn_step()
    p_lu = dev.get("step")
    dev.close()
    err = lambda x: np.abs(x.ravel() - p_ref).max() / pn
    assert err(p) <= 4 * err(p_lu) + 16 * unc + 1e-12, (err(p), err(p_lu), unc)
    ab = [np.abs(b) for b in bands]
    bwd = (np.abs(ol.penta_multiply(*bands, p) + g.ravel()) / (ol.penta_multiply(*ab, np.abs(p)) + np.abs(g.ravel()) + 1e-300)).max()
    assert bwd <= 1e-12, bwd


@pytest.mark.parametrize("name,N,band", [("acrobot", 40, 1), ("spinner", 40, 2), ("hopper", 50, 2)])   # (2: also where the default keeps a batch on the block kernels)
def test_band_in_a_batch_and_failure_report(name, N, band):
    """grid.y = problem: bit-identical to single-problem contexts; a singular Hessian is reported for its problem"""
    B = 3
    cfg, model = load_config(name), load_model(name)
    probs, qs = [], []
    for b in range(B):
        prob, sp, _ = make_problem(cfg, model, num_steps=N)
        sp.scaling = False
        sp.equality_constraints = False
        prob.q_nom = prob.q_nom + 0.01 * b
        probs.append(prob)
        qs.append(synthetic_trajectory(cfg, model, N, seed=b, lower=0.01))
    batch = hip.HipPath(model, probs, sp)
    batch.set_option("solver_band", band)
    batch.set_option("gn_small", 0)
    batch.set_q_batch(np.array(qs))
    for _ in range(2):
        batch.gn_step()
    assert batch.get_option("last_solver") == 6
    for b in range(B):
        one = hip.HipPath(model, probs[b], sp)
        one.set_option("solver_band", band)
        one.set_option("gn_small", 0)
        one.set_q(qs[b])
        one.gn_step()
        assert one.get_option("last_solver") == 6
        assert np.array_equal(batch.get("step", b), one.get("step"))
        one.close()
    bad = copy.deepcopy(probs[1])
    for Wt in (bad.Qq, bad.Qv, bad.Qf_q, bad.Qf_v):
        Wt[1, :] = 0.0
        Wt[:, 1] = 0.0
    bad.R[:] = 0.0
    batch.set_problem_batch(1, bad)
    batch.gn_step()
    assert batch.solver_status_batch() == [False, True, False]
    batch.close()
