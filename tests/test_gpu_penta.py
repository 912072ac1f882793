"""The reference's pure linear-algebra tests (optimizer/test/penta_diagonal_solver_test.cc: test
name and line cited per test) run against the DEVICE solvers through the C-ABI: the bands are
written straight into the context's Hessian storage (IDTO_ARR_HBANDS: [A | B | C], each N+6
column-major blocks, lower bands only - the solvers mirror them), the right-hand sides are
device buffers.  Both solvers are exercised: the bit-exact restatement of the reference's
pivoted-LU block Thomas ("reference_solver") and the production banded block LDL^T (one- and
two-sided, single and many right-hand sides)."""
import numpy as np
import pytest
import torch

import oracle_lib as ol
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem
from test_oracle_penta import EPS, _spd_blocks, compare, from_lower_dense

pytestmark = pytest.mark.gpu

MODEL_FOR_BLOCK = {2: "acrobot", 3: "spinner", 5: "hopper"}   # block size = nq of the model


class DeviceSolver:
    """a context of the right shape whose Hessian bands the test overwrites"""

    def __init__(self, bs, n):
        name = MODEL_FOR_BLOCK[bs]
        cfg, model = load_config(name), load_model(name)
        prob, sp, _ = make_problem(cfg, model, num_steps=n - 1)
        self.dev = hip.HipPath(model, prob, sp)
        self.bs, self.n = bs, n
        span = (n + 5) * bs * bs   # (N + 6) blocks per band

        class _Ptr:
            __cuda_array_interface__ = {"shape": (3 * span,), "typestr": "<f8",
                                        "data": (self.dev.device_ptr("hbands"), False), "version": 2}
        self.view = torch.as_tensor(_Ptr(), device="cuda:0")
        self.span = span

    def set_bands(self, A, B, C):
        host = np.zeros(3 * self.span)
        for k, band in enumerate((A, B, C)):
            # [blk, row, col] -> column-major blocks
            host[k * self.span:k * self.span + self.n * self.bs ** 2] = np.asarray(band).transpose(0, 2, 1).ravel()
        self.view.copy_(torch.from_numpy(host))
        torch.cuda.synchronize()

    def solve(self, b, reference=False, two_sided=True):
        b = np.atleast_2d(np.asarray(b, dtype=np.float64))
        rhs = torch.tensor(b, dtype=torch.float64, device="cuda")
        x = torch.zeros_like(rhs)
        self.dev.set_option("reference_solver", int(reference))
        self.dev.set_option("two_sided", int(two_sided))
        self.dev.factor_solve(rhs.data_ptr(), b.shape[0], x.data_ptr())
        self.dev.sync()
        out = x.cpu().numpy()
        return out[0] if out.shape[0] == 1 else out


def test_solve_identity():  # :109-123 (exact)
    bs, n = 3, 5
    s = DeviceSolver(bs, n)
    Z, I = np.zeros((n, bs, bs)), np.tile(np.eye(bs), (n, 1, 1))
    s.set_bands(Z, Z, I)
    b = np.linspace(-3, 12.4, n * bs)
    assert np.array_equal(s.solve(b, reference=True), b)
    assert np.array_equal(s.solve(b), b)


def test_solve_block_diagonal():  # :125-154
    bs, n = 3, 5
    size = n * bs
    B1, B2, B3, _ = _spd_blocks(np.random.default_rng(4), bs)
    Z = np.zeros((n, bs, bs))
    C = np.array([B1, B2, B3, B1, B3])
    s = DeviceSolver(bs, n)
    s.set_bands(Z, Z, C)
    Cs, D, E = ol.penta_make_symmetric(Z, Z, C)
    b = np.linspace(-3, 12.4, size)
    assert np.array_equal(s.solve(b, reference=True), ol.penta_solve(Z, Z, Cs, D, E, b))   # bit-exact restatement
    x_expected = np.linalg.solve(ol.penta_make_dense(Z, Z, Cs, D, E), b)
    assert compare(s.solve(b), x_expected, 4 * EPS * size)


def test_solve_tri_diagonal():  # :156-186
    """(the reference's matrix here is symmetric but indefinite: only the pivoted solver applies)"""
    bs, n = 3, 5
    size = n * bs
    B1, B2, B3, B4 = _spd_blocks(np.random.default_rng(5), bs)
    Zb = np.zeros((bs, bs))
    A = np.zeros((n, bs, bs))
    B = np.array([Zb, B1, B2, B3, B4])
    C = np.array([B1, B2, B3, B1, B3])
    s = DeviceSolver(bs, n)
    s.set_bands(A, B, C)
    Cs, D, E = ol.penta_make_symmetric(A, B, C)
    b = np.linspace(-3, 12.4, size)
    x = s.solve(b, reference=True)
    assert np.array_equal(x, ol.penta_solve(A, B, Cs, D, E, b))
    assert compare(x, np.linalg.solve(ol.penta_make_dense(A, B, Cs, D, E), b), 8 * EPS * size)


@pytest.mark.parametrize("two_sided", [True, False])
def test_solve_penta_diagonal(two_sided):  # :188-257
    bs, n = 2, 21
    size = n * bs
    rng = np.random.default_rng(6)
    Ar = rng.uniform(-1, 1, (size, size))
    P = np.eye(size) + Ar @ Ar.T
    H = from_lower_dense(P, n, bs)
    Hd = ol.penta_make_dense(*H)
    s = DeviceSolver(bs, n)
    s.set_bands(H[0], H[1], H[2])
    x_gt = np.linspace(-3, 12.4, size)
    b = Hd @ x_gt
    cond = np.linalg.cond(Hd)
    assert np.array_equal(s.solve(b, reference=True), ol.penta_solve(*H, b))
    x = s.solve(b, two_sided=two_sided)
    assert np.linalg.norm(x - x_gt) / np.linalg.norm(x_gt) < 50 * cond * EPS
    # many right-hand sides share one factorisation: the factorisation kernel stops after its forward pass and
    # every column (the first included) goes through the substitution kernel - same factors, a differently
    # ordered substitution than the single-column solve above
    X = s.solve(np.stack([b, 2 * b, -b, 0.5 * b]), two_sided=two_sided)
    assert np.linalg.norm(X[0] - x) / np.linalg.norm(x) < 50 * cond * EPS
    for k, f in ((0, 1.0), (1, 2.0), (2, -1.0), (3, 0.5)):
        assert np.linalg.norm(X[k] - f * x_gt) / np.linalg.norm(x_gt) < 50 * cond * EPS


def test_condition_number_sweep():  # :260-319 (prints only in the reference; here: error ~ cond * eps)
    """block penta-diagonal SPD matrices of growing condition number: H = L L^T with a lower
    block-banded L (bandwidth two blocks) whose rows are scaled over half the target range"""
    bs, n = 5, 30
    size = n * bs
    rng = np.random.default_rng(7)
    s = DeviceSolver(bs, n)
    for cond_target in (1e1, 1e4, 1e8, 1e12):
        L = np.tril(rng.uniform(-0.3, 0.3, (size, size)), -1)
        for i in range(size):
            L[i, :max(0, (i // bs - 2) * bs)] = 0.0   # keep two block sub-diagonals
        L += np.eye(size)
        L = np.logspace(0, np.log10(cond_target) / 2, size)[:, None] * L
        P = L @ L.T
        H = from_lower_dense(P, n, bs)
        Hd = ol.penta_make_dense(*H)
        assert np.abs(Hd - P).max() <= 1e-12 * np.abs(P).max()   # P is exactly banded
        s.set_bands(H[0], H[1], H[2])
        x_gt = np.linspace(-3, 12.4, size)
        cond = np.linalg.cond(Hd)
        assert cond > 0.01 * cond_target
        for kw in (dict(reference=True), dict(two_sided=True), dict(two_sided=False)):
            x = s.solve(Hd @ x_gt, **kw)
            assert np.linalg.norm(x - x_gt) / np.linalg.norm(x_gt) < 100 * cond * EPS, (cond_target, kw)


@pytest.mark.parametrize("name,N", [("hopper", 50), ("mini_cheetah", 40), ("allegro_hand", 30), ("acrobot", 40), ("spinner", 9)])
def test_many_right_hand_sides_are_as_accurate_as_one(name, N):
    """several right-hand sides: the factorisation kernel stops after its forward pass and every column goes through
    penta_apply_kernel (two wavefronts per column, one per chain of the twisted factorisation).  Against the
    extended-precision solution of the assembled Gauss-Newton system the columns must be as good as the
    single-column solve (whose substitution runs inside the factorisation kernel) and as the pivoted LU."""
    from idto_amd import hip
    from idto_amd.model import load_model
    from idto_amd.problem import load_config, make_problem, synthetic_trajectory
    cfg, model = load_config(name), load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    sp.scaling = False
    sp.equality_constraints = False
    q = synthetic_trajectory(cfg, model, N, seed=0, lower=0.01)
    dev = hip.HipPath(model, prob, sp)
    dev.set_q(q)
    dev.gn_step()
    g = dev.get("gradient").ravel()
    bands = [dev.get(k) for k in ("H_A", "H_B", "H_C")]
    Cs, Dm, Em = ol.penta_make_symmetric(*bands)
    Hd = ol.penta_make_dense(bands[0], bands[1], Cs, Dm, Em)
    ref, unc = ol.refined_solution(Hd, -g)
    scale = np.abs(ref).max()
    err_one = np.abs(dev.get("step").ravel() - ref).max() / scale
    err_lu = np.abs(np.linalg.solve(Hd, -g) - ref).max() / scale
    X = dev.solve_host(np.stack([-g, 2.0 * g, -0.5 * g]))
    for col, f in enumerate((1.0, -2.0, 0.5)):
        err = np.abs(X[col] - f * ref).max() / scale / abs(f)
        assert err <= 4 * max(err_one, err_lu) + 16 * unc + 1e-12, (col, err, err_one, err_lu, unc)
    assert dev.solver_status() == (False, 0)
    dev.close()
