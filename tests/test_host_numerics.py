"""CPU tests of the host-side numerics of libidto_opt.so that need no device: the dense pivoted
LDL^T used for the Lagrange multipliers (reference trajectory_optimizer.cc:1395 calls Eigen's
ldlt(), which pivots on the diagonal and tolerates semi-definite matrices)."""
import ctypes as C

import numpy as np
import pytest

from idto_amd import optimizer


def ldlt_solve(S, b):
    n = S.shape[0]
    M = np.asfortranarray(S, dtype=np.float64).copy(order="F")
    M[np.triu_indices(n, 1)] = 1e300  # only the lower triangle may be read
    x = np.array(b, dtype=np.float64)
    rc = optimizer.lib().idto_opt_dense_ldlt_solve(M.ctypes.data_as(C.POINTER(C.c_double)), n,
                                                   x.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0
    return x


@pytest.mark.parametrize("n", [1, 2, 5, 16, 17, 33, 40, 120, 150, 240, 360])
def test_dense_ldlt_spd(n):
    rng = np.random.default_rng(n)
    G = rng.normal(size=(n, n + 3))
    S = G @ G.T
    b = rng.normal(size=n)
    x = ldlt_solve(S, b)
    ref = np.linalg.solve(S, b)
    assert np.abs(S @ x - b).max() <= 1e-10 * (np.abs(S).max() * np.abs(x).max() + np.abs(b).max())
    assert np.abs(x - ref).max() <= 1e-7 * max(1.0, np.abs(ref).max()) * max(1.0, np.linalg.cond(S) * 1e-6)


@pytest.mark.parametrize("n,rank", [(7, 4), (40, 31), (120, 117), (240, 200)])
def test_dense_ldlt_semidefinite_consistent_system(n, rank):
    """rank-deficient S with b in its range (redundant constraints): the solve returns a solution
    of the system (Eigen's ldlt() behaviour the reference relies on), not garbage"""
    rng = np.random.default_rng(n + rank)
    G = rng.normal(size=(n, rank))
    S = G @ G.T
    b = S @ rng.normal(size=n)
    x = ldlt_solve(S, b)
    assert np.abs(S @ x - b).max() <= 1e-8 * (np.abs(S).max() * max(1.0, np.abs(x).max()) + np.abs(b).max())


def test_dense_ldlt_needs_pivoting():
    """a zero leading diagonal entry: no-pivot LDL^T would divide by zero"""
    S = np.array([[0.0, 0.0, 0.0], [0.0, 4.0, 1.0], [0.0, 1.0, 3.0]])
    b = np.array([0.0, 1.0, 2.0])
    x = ldlt_solve(S, b)
    assert np.allclose(S @ x, b, atol=1e-14) and np.all(np.isfinite(x))
