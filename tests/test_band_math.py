"""The algebra of csrc/penta_band.h in numpy (CPU; the kernel itself is tested on the device by tests/test_gpu_band.py):
a block penta-diagonal matrix with blocks of K read as a scalar symmetric band of half width w = 3 K - 1, factorised
L D L^T without pivoting by TWO chains that start at the two ends - the second one on a copy of its part that is
mirrored block by block - and meet in W = 3 K middle rows; the right-hand side rides along as the band's last row;
back substitution in push form from the middle outwards.  What is checked: the storage (16 cells per column), the
mirror map and its inverse, the join (which entry of the mirrored chain's window is which entry of the middle block),
the padding that makes the chains' lengths multiples of W, the solution against numpy's dense solve, the pivots'
signs for a KKT system (nq positive then nu negative per block - and the zero pivot a row-by-row reversal would
meet, which is why the mirror keeps the order inside a block)."""
import numpy as np
import pytest

FRONT = 32


def block_penta(n, K, rng, kkt_nu=0):
    """SPD block penta-diagonal matrix (blocks K); with kkt_nu > 0 the last kkt_nu rows of every block are multiplier
    rows of a KKT system [[H, J^T], [J, 0]] interleaved as csrc/kkt.h does: zero diagonal block, banded J"""
    M = n * K
    nq = K - kkt_nu
    A = np.zeros((M, M))
    R = rng.uniform(-1, 1, (M, M))
    S = R @ R.T + M * np.eye(M)
    for t in range(n):
        for s in range(max(0, t - 2), min(n, t + 3)):
            A[t * K:(t + 1) * K, s * K:(s + 1) * K] = S[t * K:(t + 1) * K, s * K:(s + 1) * K]
    if kkt_nu:
        for t in range(n):
            mu = slice(t * K + nq, (t + 1) * K)
            A[mu, :] = 0.0
            A[:, mu] = 0.0
        for t in range(n):
            mu = slice(t * K + nq, (t + 1) * K)
            for s in range(max(0, t - 2), t + 1):   # mu_t's row: the variables of block rows t - 2 .. t
                Jb = rng.uniform(-1, 1, (kkt_nu, nq))
                if s == t:
                    Jb += 3.0 * np.eye(kkt_nu, nq)   # full row rank against the block's own variables
                A[mu, s * K:s * K + nq] = Jb
                A[s * K:s * K + nq, mu] = Jb.T
    return A


def mirror(i, M, K):
    t = i // K
    return (M // K - 1 - t) * K + (i - t * K)


def layout(M, W):
    m = ((M - W) // 2 + W // 2) // W * W
    lim, nb = m + W, M - W - m
    pad = (W - nb % W) % W
    return m, lim, nb, pad


def make_copies(A, b, K):
    """the two chains' copies: [column][16 cells]; cell d = entry (q + d, q) of the chain's own ordering, cell 15 = rhs"""
    M, W = A.shape[0], 3 * K
    w = W - 1
    m, lim, nb, pad = layout(M, W)
    T = np.zeros((FRONT + lim + 2 * W + 1, 16))
    Bm = np.zeros((FRONT + pad + nb + 2 * W + 1, 16))
    for i in range(lim):
        for d in range(w + 1):
            if i + d < lim:
                T[FRONT + i, d] = A[i + d, i]
        T[FRONT + i, 15] = b[i]
    for i in range(lim, lim + 2 * W + 1):
        T[FRONT + i, 0] = 1.0
    for c in range(pad):
        Bm[FRONT + c, 0] = 1.0
    for i in range(nb):
        for d in range(w + 1):
            if i + d < M:
                Bm[FRONT + pad + i, d] = A[mirror(i + d, M, K), mirror(i, M, K)]
        Bm[FRONT + pad + i, 15] = b[mirror(i, M, K)]
    return T, Bm


def forward(arr, dinv, W, g0, g1, R):
    """groups of W pivots on a copy; R: the window [slot][16 lanes] (in / out).  Lane 15 is the right-hand side."""
    w = W - 1
    for g in range(g0, g1):
        for S in range(W):
            j = g * W + S
            col = R[S].copy()
            inv = 1.0 / col[0]
            lp = col * inv
            arr[FRONT + j, :] = lp
            dinv[j] = inv
            l0, ly = lp.copy(), np.zeros(16)
            l0[15], ly[15] = 0.0, lp[15]
            for c in range(1, W):
                lsh = np.zeros(16)
                lsh[:16 - c] = l0[c:]          # row_shl:c with zero fill
                R[(S + c) % W] -= (lsh + ly) * col[c]
            R[S] = arr[FRONT + j + W].copy()


def backward(arr, j1, j0, P, W):
    w = W - 1
    for j in range(j1 - 1, j0 - 1, -1):
        Lr = np.zeros(16)
        for d in range(1, w + 1):
            Lr[d] = arr[FRONT + j - d, d]
        xj = arr[FRONT + j, 15] + P[0]
        arr[FRONT + j, 15] = xj
        P = P - Lr * xj
        P = np.append(P[1:], 0.0)
    return P


def band_solve(A, b, K):
    M, W = A.shape[0], 3 * K
    w = W - 1
    m, lim, nb, pad = layout(M, W)
    T, Bm = make_copies(A, b, K)
    Dt, Db = np.zeros(T.shape[0]), np.zeros(Bm.shape[0])
    # first chain to the join
    R1 = np.array([T[FRONT + S].copy() for S in range(W)])
    forward(T, Dt, W, 0, m // W, R1)
    # mirrored chain: all of its pivots (pad identity ones first)
    R2 = np.array([Bm[FRONT + S].copy() for S in range(W)])
    forward(Bm, Db, W, 0, (nb + pad) // W, R2)
    # the join: row m + x of the middle is the mirrored chain's index nb + u(x)
    um = lambda x: (2 - x // K) * K + x % K
    for S in range(W):
        for lane in range(16):
            d = 0 if lane == 15 else lane
            if S + d < W and d <= w:
                ua, ub = um(S + d), um(S)
                u1, u2 = max(ua, ub), min(ua, ub)
                R1[S, lane] += R2[u2, 15 if lane == 15 else u1 - u2]
    forward(T, Dt, W, m // W, m // W + 1, R1)
    # back substitution: the middle, then both chains
    P = backward(T, lim, m, np.zeros(16), W)
    backward(T, m, 0, P, W)
    P = np.zeros(16)
    for c in range(W - 1, -1, -1):
        j = nb + pad + c
        xj = T[FRONT + mirror(nb + c, M, K), 15]
        Lr = np.zeros(16)
        for d in range(1, w + 1):
            Lr[d] = Bm[FRONT + j - d, d]
        P = np.append((P - Lr * xj)[1:], 0.0)
    backward(Bm, nb + pad, pad, P, W)
    x = np.array([T[FRONT + j, 15] if j < lim else Bm[FRONT + pad + mirror(j, M, K), 15] for j in range(M)])
    d = np.array([1.0 / (Dt[j] if j < lim else Db[pad + mirror(j, M, K)]) for j in range(M)])
    return x, d


@pytest.mark.parametrize("K", [2, 3, 4, 5])
@pytest.mark.parametrize("n", [12, 24, 41, 50])
def test_two_chains_solve_the_system(K, n):
    rng = np.random.default_rng(10 * K + n)
    A = block_penta(n, K, rng)
    b = rng.uniform(-1, 1, n * K)
    x, d = band_solve(A, b, K)
    assert np.all(d > 0)
    assert np.allclose(x, np.linalg.solve(A, b), rtol=0, atol=1e-11 * np.abs(x).max())


@pytest.mark.parametrize("nq,nu", [(2, 1), (3, 1), (2, 2)])
def test_kkt_system_keeps_its_pivot_pattern_on_both_chains(nq, nu):
    """nq positive then nu negative pivots per block, on the mirrored chain too (csrc/kkt.h)"""
    K, n = nq + nu, 30
    rng = np.random.default_rng(nq * 7 + nu)
    A = block_penta(n, K, rng, kkt_nu=nu)
    for t in range(n):   # (block row 0 of the real system is decoupled; here every multiplier row has its variables)
        assert np.all(np.diag(A)[t * K + nq:(t + 1) * K] == 0.0)
    b = rng.uniform(-1, 1, n * K)
    x, d = band_solve(A, b, K)
    assert np.allclose(x, np.linalg.solve(A, b), rtol=0, atol=1e-9 * np.abs(x).max())
    d = d.reshape(n, K)
    assert np.all(d[:, :nq] > 0) and np.all(d[:, nq:] < 0)


def test_row_by_row_reversal_would_meet_a_zero_pivot():
    """why the mirror keeps the order inside a block: reversed row by row, the LAST block's multiplier row comes first,
    and its diagonal entry is zero"""
    nq, nu, n = 2, 1, 24
    K = nq + nu
    A = block_penta(n, K, np.random.default_rng(5), kkt_nu=nu)
    M = n * K
    rev = A[::-1, ::-1]
    assert rev[0, 0] == 0.0
    mir = np.array([mirror(i, M, K) for i in range(M)])
    Am = A[np.ix_(mir, mir)]
    assert Am[0, 0] > 0.0
    assert np.array_equal(mir[mir], np.arange(M))                       # its own inverse
    i, j = np.nonzero(Am)
    assert np.abs(i - j).max() <= 3 * K - 1                             # the half width stays


@pytest.mark.parametrize("K,n", [(2, 40), (3, 40), (5, 50), (2, 200)])
def test_layout(K, n):
    M, W = n * K, 3 * K
    m, lim, nb, pad = layout(M, W)
    assert m % W == 0 and m >= W and nb >= 1 and (nb + pad) % W == 0 and m + W + nb == M
    assert abs(m - nb) <= W   # the two chains are as long as each other up to a group
