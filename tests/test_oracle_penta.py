"""Pins the oracle's block penta-diagonal algebra (oracle/penta.h) with the cases of
the reference's optimizer/test/penta_diagonal_solver_test.cc (test name and line
cited per test).  Eigen is not available, so the dense reference solutions come
from numpy.linalg (LAPACK) instead of `Hdense.ldlt().solve(b)`."""
import numpy as np

import oracle_lib as ol

EPS = np.finfo(float).eps


def compare(a, b, tol):
    """drake::CompareMatrices(..., relative): |a-b| <= tol*max(1,|a|,|b|)
    (reference utils/eigen_matrix_compare.h:95-98)."""
    a, b = np.asarray(a), np.asarray(b)
    return bool(np.all(np.abs(a - b) <= tol * np.maximum(1.0, np.maximum(np.abs(a), np.abs(b)))))


def from_lower_dense(M, n, bs):
    """PentaDiagonalMatrix::MakeSymmetricFromLowerDense (penta_diagonal_matrix.cc:130-146)."""
    A, B, C = np.zeros((n, bs, bs)), np.zeros((n, bs, bs)), np.zeros((n, bs, bs))
    for i in range(n):
        if i >= 2:
            A[i] = M[i * bs:(i + 1) * bs, (i - 2) * bs:(i - 1) * bs]
        if i >= 1:
            B[i] = M[i * bs:(i + 1) * bs, (i - 1) * bs:i * bs]
        C[i] = M[i * bs:(i + 1) * bs, i * bs:(i + 1) * bs]
    Cs, D, E = ol.penta_make_symmetric(A, B, C)
    return A, B, Cs, D, E


def test_multiply_by():  # :20-40
    bs, n = 2, 5
    size = n * bs
    rng = np.random.default_rng(0)
    H = from_lower_dense(rng.uniform(-1, 1, (size, size)), n, bs)
    v = np.linspace(0.1, 1.1, size)
    assert compare(ol.penta_multiply(*H, v), ol.penta_make_dense(*H) @ v, EPS * size)


def test_symmetric_matrix():  # :88-107
    k = 5
    Z = np.zeros((k, k))
    B1, B2, B3, B4, B5, B6 = [c * np.ones((k, k)) for c in (1.5, 2.1, -12.8, 1.8, 15.3, 7.1)]
    A, B, C = np.array([Z, Z, B1]), np.array([Z, B2, B3]), np.array([B4, B5, B6])
    Cs, D, E = ol.penta_make_symmetric(A, B, C)
    assert np.array_equal(D[0], B[1].T) and np.array_equal(D[1], B[2].T) and np.array_equal(D[2], Z)
    assert np.array_equal(E[0], A[2].T) and np.array_equal(E[1], Z) and np.array_equal(E[2], Z)
    dense = ol.penta_make_dense(A, B, Cs, D, E)
    assert np.array_equal(dense, dense.T)


def test_make_symmetric_uses_lower_triangle_of_C():  # penta_diagonal_matrix.cc:71-76
    rng = np.random.default_rng(3)
    C = rng.uniform(-1, 1, (3, 4, 4))
    Z = np.zeros((3, 4, 4))
    Cs, _, _ = ol.penta_make_symmetric(Z, Z, C)
    for i in range(3):
        assert np.array_equal(np.tril(Cs[i]), np.tril(C[i]))
        assert np.array_equal(Cs[i], Cs[i].T)


def test_solve_identity():  # :109-123  (exact)
    bs, n = 3, 5
    I = np.tile(np.eye(bs), (n, 1, 1))
    Z = np.zeros((n, bs, bs))
    b = np.linspace(-3, 12.4, n * bs)
    x = ol.penta_solve(Z, Z, I, Z, Z, b)
    assert np.array_equal(x, b)


def _spd_blocks(rng, bs):
    R = rng.uniform(-1, 1, (bs, bs))
    I = np.eye(bs)
    return [c * I + R @ R.T for c in (2.1, 3.5, 0.2, 1.3)]


def test_solve_block_diagonal():  # :125-154
    bs, n = 3, 5
    size = n * bs
    B1, B2, B3, _ = _spd_blocks(np.random.default_rng(4), bs)
    Z = np.zeros((n, bs, bs))
    C = np.array([B1, B2, B3, B1, B3])
    Cs, D, E = ol.penta_make_symmetric(Z, Z, C)
    b = np.linspace(-3, 12.4, size)
    x = ol.penta_solve(Z, Z, Cs, D, E, b)
    x_expected = np.linalg.solve(ol.penta_make_dense(Z, Z, Cs, D, E), b)
    assert compare(x, x_expected, EPS * size)


def test_solve_tri_diagonal():  # :156-186
    bs, n = 3, 5
    size = n * bs
    B1, B2, B3, B4 = _spd_blocks(np.random.default_rng(5), bs)
    Zb = np.zeros((bs, bs))
    A = np.zeros((n, bs, bs))
    B = np.array([Zb, B1, B2, B3, B4])
    C = np.array([B1, B2, B3, B1, B3])
    Cs, D, E = ol.penta_make_symmetric(A, B, C)
    b = np.linspace(-3, 12.4, size)
    x = ol.penta_solve(A, B, Cs, D, E, b)
    x_expected = np.linalg.solve(ol.penta_make_dense(A, B, Cs, D, E), b)
    # the reference asserts eps*size against Eigen's LDLT; LAPACK's LU differs from
    # both by a few ulp more on this (indefinite, cond ~1e2) matrix
    assert compare(x, x_expected, 8 * EPS * size)


def test_solve_penta_diagonal():  # :188-257 (the reference only asserts status == success)
    bs, n = 2, 21
    size = n * bs
    rng = np.random.default_rng(6)
    Ar = rng.uniform(-1, 1, (size, size))
    P = np.eye(size) + Ar @ Ar.T
    H = from_lower_dense(P, n, bs)
    Hd = ol.penta_make_dense(*H)
    x_gt = np.linspace(-3, 12.4, size)
    b = Hd @ x_gt
    assert compare(ol.penta_multiply(*H, x_gt), b, EPS * size * np.abs(b).max())
    x = ol.penta_solve(*H, b)
    cond = np.linalg.cond(Hd)
    assert np.linalg.norm(x - x_gt) / np.linalg.norm(x_gt) < 50 * cond * EPS
    # multiple right-hand sides share one factorisation
    X = ol.penta_solve(*H, np.stack([b, 2 * b, -b]))
    assert np.array_equal(X[0], x) and compare(X[1], 2 * x, 1e-12) and np.array_equal(X[2], -x)


def test_condition_number_sweep():  # :260-319 (prints only in the reference; here: error ~ cond*eps)
    bs, n = 5, 30
    size = n * bs
    rng = np.random.default_rng(7)
    for cond_target in (1e1, 1e4, 1e8, 1e12):
        Q, _ = np.linalg.qr(rng.normal(size=(size, size)))
        ev = np.logspace(0, np.log10(cond_target), size)
        P = (Q * ev) @ Q.T
        H = from_lower_dense(P, n, bs)
        Hd = ol.penta_make_dense(*H)
        if np.linalg.eigvalsh(Hd).min() <= 0:
            continue  # truncating to the band can destroy positive definiteness
        x_gt = np.linspace(-3, 12.4, size)
        x = ol.penta_solve(*H, Hd @ x_gt)
        assert np.linalg.norm(x - x_gt) / np.linalg.norm(x_gt) < 100 * np.linalg.cond(Hd) * EPS


def test_extract_and_scale_by_diagonal():  # :321-371
    bs, n = 3, 4
    size = n * bs
    rng = np.random.default_rng(8)
    Ar = rng.uniform(-1, 1, (size, size))
    H = from_lower_dense(3 * np.eye(size) + Ar @ Ar.T, n, bs)
    Hd = ol.penta_make_dense(*H)
    s = rng.uniform(0.5, 2.0, size)
    Hs = ol.penta_scale_by_diagonal(*H, s)
    assert compare(ol.penta_make_dense(*Hs), np.diag(s) @ Hd @ np.diag(s), EPS)
    assert np.array_equal(np.diag(Hd), np.concatenate([np.diag(c) for c in H[2]]))


def test_refined_solution_against_mpmath():
    """oracle_lib.refined_solution (the extended-precision yardstick of the GPU solver parity tests)
    reproduces a 60-digit mpmath solve to double rounding on SPD systems up to cond ~ 1e12"""
    import mpmath as mp
    rng = np.random.default_rng(7)
    size = 40
    for cond_target in (1e4, 1e8, 1e12):
        L = np.tril(rng.uniform(-0.3, 0.3, (size, size)), -1)
        for i in range(size):
            L[i, :max(0, i - 10)] = 0.0
        L += np.eye(size)
        L = np.logspace(0, np.log10(cond_target) / 2, size)[:, None] * L
        H = L @ L.T
        H = (H + H.T) / 2
        b = rng.normal(size=size)
        x, unc = ol.refined_solution(H, b)
        mp.mp.dps = 60
        xm = np.array([float(v) for v in mp.lu_solve(mp.matrix(H.tolist()), mp.matrix(b.tolist()))])
        assert np.abs(x - xm).max() <= 4 * EPS * np.abs(xm).max()
        assert unc < 1e-15
        # and a plain double solve is measurably worse on the ill-conditioned one
        if cond_target >= 1e12:
            assert np.abs(np.linalg.solve(H, b) - xm).max() > 100 * np.abs(x - xm).max()
