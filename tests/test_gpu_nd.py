"""Nested-dissection form of the block LDL^T solver - csrc/penta_pipe.h (production for block sizes up to 20:
pipelined chains, five workgroups, the joiners carry their coupling to the separator; `last_solver` 4) and
csrc/penta_nd.h (two producer / joiner chain pairs around a separator, two spike workgroups carrying the
chains' coupling to the separator; `last_solver` 2) - against (i) the two-workgroup form of the same factorisation, (ii) the bit-exact restatement of the
reference's pivoted-LU block Thomas (optimizer/penta_diagonal_solver.h:124-248) and (iii) an
extended-precision solution: same accuracy bar as tests/test_gpu_parity.py, repeated launches
bit-identical (the workgroups synchronise through device-memory flags and counters), batches,
failure reporting, and the reference's own penta-diagonal test matrix."""
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


@pytest.fixture(autouse=True)
def _block_kernels_only(monkeypatch):
    """these tests are about the multi-workgroup block kernels: the scalar band factorisation (penta_band.h, tests/test_gpu_band.py), which takes the small models' systems by default, stays out"""
    monkeypatch.setenv("IDTO_SOLVER_BAND", "0")


def _setup(name, N, seed=0):
    cfg, model = load_config(name), load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    sp.scaling = False
    sp.equality_constraints = False
    return cfg, model, prob, sp, synthetic_trajectory(cfg, model, N, seed=seed, lower=0.01)


@pytest.mark.parametrize("name,N", [("mini_cheetah", 40), ("mini_cheetah", 24), ("mini_cheetah", 31), ("hopper", 50),
                                    ("spinner", 40), ("acrobot", 40), ("acrobot", 63), ("allegro_hand", 60),
                                    ("allegro_hand", 27)])
@pytest.mark.parametrize("pipe", [1, 0])
def test_nested_dissection_solver(name, N, pipe):
    cfg, model, prob, sp, q = _setup(name, N)
    dev = hip.HipPath(model, prob, sp)
    dev.set_option("solver_pipe", pipe)
    dev.set_q(q)
    dev.gn_step()
    want = 4 if (pipe and model.nq <= 20) else 2
    assert dev.get_option("last_solver") == want, "the nested-dissection kernel was expected to take this size"
    p_nd = dev.get("step")
    for _ in range(3):   # flags / counters are epoch-valued: repeated launches must reproduce the bits
        dev.factor_solve()
        assert np.array_equal(dev.get("step"), p_nd)
    assert dev.solver_status() == (False, 0)
    dev.set_option("solver_nd", 0)
    dev.factor_solve()
    assert dev.get_option("last_solver") == 1
    p_two = dev.get("step")
    dev.set_option("reference_solver", 1)
    dev.factor_solve()
    p_lu = dev.get("step")
    orc = Oracle(model, prob, sp)
    g, bands = orc.grad_hess(q)
    assert np.array_equal(dev.get("gradient"), g)
    p_ref, unc = ol.refined_solution(ol.penta_make_dense(*bands), -g.ravel())
    pn = np.abs(p_ref).max()
    err = lambda x: np.abs(x.ravel() - p_ref).max() / pn
    # componentwise backward error |H x + g| / (|H| |x| + |g|).  The pipelined solver's back substitution multiplies
    # by precomputed U^-1 blocks (penta_pipe.h pipe_backward): ~1e-14 where the row-by-row substitution has ~1e-16 and
    # the reference's pivoted-LU block Thomas ~1e-14 .. 1e-12 (tools/nd_accuracy.py); the forward error is what the
    # Gauss-Newton step sees and is bounded against LU's above.
    ab = [np.abs(b) for b in bands]
    bwd = lambda x: (np.abs(ol.penta_multiply(*bands, x) + g.ravel()) /
                     (ol.penta_multiply(*ab, np.abs(x)) + np.abs(g.ravel()) + 1e-300)).max()
    assert err(p_nd) <= 4 * err(p_lu) + 16 * unc + 1e-12, (err(p_nd), err(p_two), err(p_lu))
    assert bwd(p_nd) <= 16 * bwd(p_lu) + 2e-13, (bwd(p_nd), bwd(p_lu))
    dev.close()


@pytest.mark.parametrize("N", [24, 25, 30, 33, 40, 47, 54, 60])
def test_seven_workgroup_back_substitution_in_recursion_form_over_the_horizons(N):
    """penta_pipe.h chain_recursion_tail: allegro's 23 x 23 blocks on the seven-workgroup kernel with the
    back substitution in recursion form (W in global memory, formed by the pair's producer) against the row-by-row
    tail of the same kernel and the bit-exact restatement of the reference's solver, at horizons that give the four
    chains every mix of lengths (even / odd halves, 5 .. 16 rows)."""
    cfg, model, prob, sp, q = _setup("allegro_hand", N)
    dev = hip.HipPath(model, prob, sp)
    dev.set_q(q)
    steps = {}
    for rec in (1, 0):
        dev.set_option("nd_recursion", rec)
        dev.gn_step()
        assert dev.get_option("last_solver") == 2 and dev.solver_status() == (False, 0)
        steps[rec] = dev.get("step")
        for _ in range(2):   # (epoch-valued flags and counters: the bits repeat)
            dev.factor_solve()
            assert np.array_equal(dev.get("step"), steps[rec])
    dev.set_option("reference_solver", 1)
    dev.factor_solve()
    p_lu = dev.get("step")
    orc = Oracle(model, prob, sp)
    g, bands = orc.grad_hess(q)
    p_ref, unc = ol.refined_solution(ol.penta_make_dense(*bands), -g.ravel())
    pn = np.abs(p_ref).max()
    err = lambda x: np.abs(x.ravel() - p_ref).max() / pn
    assert not np.array_equal(steps[1], steps[0]), "the option did not change the tail: the case tests nothing"
    for rec in (1, 0):
        assert err(steps[rec]) <= 4 * err(p_lu) + 16 * unc + 1e-12, (rec, err(steps[rec]), err(p_lu))
    dev.close()


def test_reference_penta_diagonal_case_through_nd():
    """penta_diagonal_solver_test.cc:188-257 (SPD block penta-diagonal system, known solution) at a
    size the nested-dissection kernel takes (bands written into the context, explicit right-hand side)"""
    bs, n = 2, 41
    size = n * bs
    rng = np.random.default_rng(6)
    Ar = rng.uniform(-1, 1, (size, size))
    H = from_lower_dense(np.eye(size) + Ar @ Ar.T, n, bs)
    Hd = ol.penta_make_dense(*H)
    s = DeviceSolver(bs, n)
    s.set_bands(H[0], H[1], H[2])
    x_gt = np.linspace(-3, 12.4, size)
    x = s.solve(Hd @ x_gt)
    assert s.dev.get_option("last_solver") == 4
    assert np.linalg.norm(x - x_gt) / np.linalg.norm(x_gt) < 50 * np.linalg.cond(Hd) * np.finfo(float).eps
    # many right-hand sides: the two-workgroup factors serve the substitution kernel
    X = s.solve(np.stack([Hd @ x_gt, 2 * (Hd @ x_gt)]))
    assert s.dev.get_option("last_solver") == 1
    assert np.linalg.norm(X[1] - 2 * x_gt) / np.linalg.norm(x_gt) < 100 * np.linalg.cond(Hd) * np.finfo(float).eps


@pytest.mark.parametrize("pipe", [1, 0])
def test_nd_in_a_batch_and_failure_report(pipe):
    name, N, B = "mini_cheetah", 40, 3
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
    batch.set_option("solver_pipe", pipe)
    batch.set_q_batch(np.array(qs))
    batch.gn_step()
    assert batch.get_option("last_solver") == (4 if pipe else 2)
    for b in range(B):
        one = hip.HipPath(model, probs[b], sp)
        one.set_option("solver_pipe", pipe)
        one.set_q(qs[b])
        one.gn_step()
        assert np.array_equal(batch.get("step", b), one.get("step"))
        one.close()
    # a singular Hessian in one problem of the batch: reported for that problem by whichever workgroup meets it
    import copy
    bad = copy.deepcopy(probs[1])
    for W in (bad.Qq, bad.Qv, bad.Qf_q, bad.Qf_v):
        W[7, :] = 0.0
        W[:, 7] = 0.0
    bad.R[:] = 0.0
    batch.set_problem_batch(1, bad)
    batch.gn_step()
    assert batch.solver_status_batch() == [False, True, False]
    batch.close()


@pytest.mark.parametrize("pipe", [1, 0])
def test_changing_inputs_every_launch(pipe):
    """the trajectory changes every launch: a read that is not ordered after its producer would return
    the previous launch's values (tools/nd_stress.py is the long version)"""
    cfg, model, prob, sp, _ = _setup("mini_cheetah", 40)
    qs = [synthetic_trajectory(cfg, model, 40, seed=s, lower=0.01) for s in range(3)]
    dev = hip.HipPath(model, prob, sp)
    dev.set_option("solver_nd", 0)
    ref = []
    for q in qs:
        dev.set_q(q)
        dev.gn_step()
        ref.append(dev.get("step"))
    dev.set_option("solver_nd", 1)
    dev.set_option("solver_pipe", pipe)
    for it in range(60):
        j = it % 3
        dev.set_q(qs[j])
        dev.gn_step()
        assert dev.get_option("last_solver") == (4 if pipe else 2)
        p = dev.get("step")
        assert np.abs(p - ref[j]).max() <= 1e-3 * np.abs(ref[j]).max(), it   # (another launch's data would be off by O(1); two factorisations differ ~cond * eps)
    dev.close()


def test_many_contexts_on_concurrent_streams():
    """the solver's workgroups wait for each other through device-memory flags in a plain
    (non-cooperative) launch.  Many contexts, each on its own stream and host thread, iterate at
    once (seven workgroups per nested-dissection launch, one CU each): every launch must complete
    and give its own problem's step."""
    import threading
    cfg, model, prob, sp, _ = _setup("mini_cheetah", 40)
    n_ctx, iters = 24, 40
    qs = [synthetic_trajectory(cfg, model, 40, seed=s, lower=0.01) for s in range(n_ctx)]
    devs = [hip.HipPath(model, prob, sp) for _ in range(n_ctx)]
    ref = []
    for d, q in zip(devs, qs):
        d.set_q(q)
        d.gn_step()
        ref.append(d.get("step"))
    gate = threading.Barrier(n_ctx)
    errors = []

    def worker(i):
        try:
            gate.wait()
            for _ in range(iters):
                devs[i].gn_step()
            devs[i].sync()
            if not np.array_equal(devs[i].get("step"), ref[i]):
                errors.append((i, "step differs"))
        except Exception as e:   # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(n_ctx)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a launch did not complete (workgroups waiting on each other)"
    assert not errors, errors[:3]
    for d in devs:
        d.close()


@pytest.mark.parametrize("name,N,want", [("hopper", 150, 1), ("acrobot", 200, 1), ("mini_cheetah", 100, 4), ("hopper", 100, 4),
                                         ("mini_cheetah", 70, 4), ("spinner", 126, 4), ("spinner", 140, 1)])
def test_long_horizons(name, N, want):
    """The chains' per-row tables hold ND_MAXROWS = 32 local rows: longer horizons must fall back to the two-workgroup
    factorisation (last_solver 1) instead of failing, and the pipelined kernel's longest chains (its back
    substitution then no longer fits the LDS in recursion form at K = 19 and goes row by row) must stay accurate."""
    cfg, model, prob, sp, q = _setup(name, N)
    dev = hip.HipPath(model, prob, sp)
    dev.set_option("fused", 0)   # (the small models' single-launch iteration has its own tests: this one is about the chains)
    dev.set_q(q)
    dev.gn_step()
    assert dev.get_option("last_solver") == want
    p_fast = dev.get("step")
    assert dev.solver_status() == (False, 0)
    dev.set_option("reference_solver", 1)
    dev.factor_solve()
    p_lu = dev.get("step")
    orc = Oracle(model, prob, sp)
    g, bands = orc.grad_hess(q)
    p_ref, unc = ol.refined_solution(ol.penta_make_dense(*bands), -g.ravel())
    pn = np.abs(p_ref).max()
    err = lambda x: np.abs(x.ravel() - p_ref).max() / pn
    assert err(p_fast) <= 4 * err(p_lu) + 16 * unc + 1e-12, (err(p_fast), err(p_lu), unc)
    dev.close()


@pytest.mark.parametrize("name,N", [("mini_cheetah", 20), ("mini_cheetah", 16), ("mini_cheetah", 17), ("hopper", 20), ("spinner", 16)])
def test_pipelined_solver_takes_mpc_horizons(name, N):
    """Round 5: systems from 16 block rows on take the five-workgroup kernel (option "nd_min_rows"): the MPC examples
    plan over 20 steps (reference examples/mini_cheetah/mini_cheetah.yaml: num_steps 20), where the one-launch iteration
    cost 99 us a re-plan.  Same criteria as test_nested_dissection_solver: reproducible bits, a clean status, forward
    error within 4x the pivoted LU's against the extended-precision solution, componentwise backward error."""
    cfg, model, prob, sp, q = _setup(name, N)
    dev = hip.HipPath(model, prob, sp)
    dev.set_q(q)
    dev.gn_step()
    assert dev.get_option("last_solver") == 4 and dev.get_option("nd_min_rows") == 16
    p_nd = dev.get("step")
    for _ in range(3):
        dev.gn_step()
        assert np.array_equal(dev.get("step"), p_nd)
    assert dev.solver_status() == (False, 0)
    dev.set_option("nd_min_rows", 24)        # what rounds 1 - 4 ran at this size
    dev.gn_step()
    assert dev.get_option("last_solver") != 4
    p_before = dev.get("step")
    dev.set_option("reference_solver", 1)
    dev.factor_solve()
    p_lu = dev.get("step")
    orc = Oracle(model, prob, sp)
    g, bands = orc.grad_hess(q)
    assert np.array_equal(dev.get("gradient"), g)
    p_ref, unc = ol.refined_solution(ol.penta_make_dense(*bands), -g.ravel())
    pn = np.abs(p_ref).max()
    err = lambda x: np.abs(x.ravel() - p_ref).max() / pn
    ab = [np.abs(b) for b in bands]
    bwd = lambda x: (np.abs(ol.penta_multiply(*bands, x) + g.ravel()) /
                     (ol.penta_multiply(*ab, np.abs(x)) + np.abs(g.ravel()) + 1e-300)).max()
    assert err(p_nd) <= 4 * err(p_lu) + 16 * unc + 1e-12, (err(p_nd), err(p_before), err(p_lu))
    assert bwd(p_nd) <= 16 * bwd(p_lu) + 2e-13, (bwd(p_nd), bwd(p_lu))
    dev.close()
