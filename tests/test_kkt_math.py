"""The algebra behind csrc/kkt.h, in numpy (no GPU): the equality-constrained step of the reference
(optimizer/trajectory_optimizer.cc:1371-1396, :2139-2149: lambda = (J H^-1 J^T)^-1 (h - J H^-1 g), w = H^-1 (g + J^T lambda))
is the solution of the KKT system [H J^T; J 0] [w; -lambda] = [g; h]; with the unknowns interleaved as
z_t = [x_t; mu_t], mu_t = -lambda_{t-1}, that system is block penta-diagonal with blocks of nq + nu, and an UNPIVOTED
LDL^T of it meets nq positive and then nu negative pivots in every block - what the device's banded solvers rely on."""
import numpy as np


def _system(N=9, nq=3, nu=2, seed=0):
    rng = np.random.default_rng(seed)
    n = (N + 1) * nq
    # H: symmetric positive definite, block penta-diagonal, row 0 decoupled (q_0 is no variable)
    R = np.zeros((n + 2 * nq, n))
    for t in range(N + 1):
        for s in range(max(0, t - 2), t + 1):
            R[t * nq:(t + 1) * nq, s * nq:(s + 1) * nq] = rng.standard_normal((nq, nq))
    H = R.T @ R + 0.5 * np.eye(n)
    for t in range(N + 1):
        for s in range(N + 1):
            if abs(t - s) > 2:
                H[t * nq:(t + 1) * nq, s * nq:(s + 1) * nq] = 0.0
    H[:nq, :] = 0.0; H[:, :nq] = 0.0; H[:nq, :nq] = np.eye(nq)
    H = 0.5 * (H + H.T) + 3.0 * np.eye(n)
    H[:nq, :nq] = np.eye(nq)
    assert np.linalg.eigvalsh(H).min() > 0
    # J: row (t, j) = d tau_t[dof_j] / d q, non-zero in the columns of q_{t-1}, q_t, q_{t+1}; none in q_0's
    J = np.zeros((N * nu, n))
    for t in range(N):
        for s in (t - 1, t, t + 1):
            if s >= 1:
                J[t * nu:(t + 1) * nu, s * nq:(s + 1) * nq] = rng.standard_normal((nu, nq)) * (5.0 if s == t + 1 else 1.0)
    g = rng.standard_normal(n); g[:nq] = 0.0
    h = rng.standard_normal(N * nu)
    return H, J, g, h


def _interleaved(H, J, g, h, nq, nu, shift):
    """bands of the KKT matrix in the ordering z_t = [x_t ; mu_t], mu_t = multipliers of tau_{t - shift}"""
    n = H.shape[0]; N = n // nq - 1; K = nq + nu
    M = np.zeros(((N + 1) * K, (N + 1) * K)); b = np.zeros((N + 1) * K)
    xi = lambda t: np.arange(t * K, t * K + nq)
    for t in range(N + 1):
        for s in range(N + 1):
            M[np.ix_(xi(t), xi(s))] = H[t * nq:(t + 1) * nq, s * nq:(s + 1) * nq]
        b[xi(t)] = g[t * nq:(t + 1) * nq]
    for t in range(N + 1):
        c = t - shift   # constraint index of mu_t
        mu = np.arange(t * K + nq, (t + 1) * K)
        if 0 <= c < N:
            for s in range(N + 1):
                M[np.ix_(mu, xi(s))] = J[c * nu:(c + 1) * nu, s * nq:(s + 1) * nq]
                M[np.ix_(xi(s), mu)] = J[c * nu:(c + 1) * nu, s * nq:(s + 1) * nq].T
            b[mu] = h[c * nu:(c + 1) * nu]
        else:
            M[np.ix_(mu, mu)] = np.eye(nu)   # a dummy
    return M, b


def _ldl_unpivoted(M):
    A = M.copy(); n = A.shape[0]; d = np.zeros(n)
    for j in range(n):
        d[j] = A[j, j]
        if d[j] == 0.0:
            return d, j
        l = A[j + 1:, j] / d[j]
        A[j + 1:, j + 1:] -= np.outer(l, A[j, j + 1:])
    return d, -1


def test_interleaved_kkt_is_block_penta_diagonal_with_the_sign_pattern_and_the_reference_solution():
    for seed, (N, nq, nu) in enumerate([(9, 3, 2), (12, 5, 3), (7, 2, 1)]):
        H, J, g, h = _system(N, nq, nu, seed)
        K = nq + nu
        M, b = _interleaved(H, J, g, h, nq, nu, shift=1)
        # block penta-diagonal
        for t in range(N + 1):
            for s in range(N + 1):
                if abs(t - s) > 2:
                    assert not M[t * K:(t + 1) * K, s * K:(s + 1) * K].any()
        # unpivoted LDL^T: nq positive, nu negative pivots per block (block 0: the dummy's are +1)
        d, fail = _ldl_unpivoted(M)
        assert fail < 0
        for t in range(1, N + 1):
            assert (d[t * K:t * K + nq] > 0).all() and (d[t * K + nq:(t + 1) * K] < 0).all(), (t, d[t * K:(t + 1) * K])
        # the solution is the reference's
        z = np.linalg.solve(M, b)
        Hi = np.linalg.inv(H)
        lam = np.linalg.solve(J @ Hi @ J.T, h - J @ Hi @ g)
        w = Hi @ (g + J.T @ lam)
        for t in range(N + 1):
            assert np.allclose(z[t * K:t * K + nq], w[t * nq:(t + 1) * nq], rtol=1e-9, atol=1e-9)
            if t >= 1:
                assert np.allclose(z[t * K + nq:(t + 1) * K], -lam[(t - 1) * nu:t * nu], rtol=1e-9, atol=1e-9)


def test_multipliers_in_the_block_of_their_own_time_step_meet_a_zero_pivot():
    """mu_t = multipliers of tau_t (shift 0): tau_0 depends on q_1 only (q_0 is no variable), so its multipliers' Schur
    complement is exactly zero when block 0 is eliminated - the ordering of kkt.h (shift 1) is not a matter of taste"""
    H, J, g, h = _system(9, 3, 2, 4)
    M, _ = _interleaved(H, J, g, h, 3, 2, shift=0)
    d, fail = _ldl_unpivoted(M)
    assert fail == 3   # the first multiplier row of block 0
