"""The reference's known-answer tests (optimizer/test/trajectory_optimizer_test.cc =
"TO_test.cc"; optimizer/test/penta_diagonal_solver_test.cc) run against the HIP path
through the C-ABI: the device results are held to the closed-form values the reference
asserts, not just to the oracle."""
import numpy as np
import pytest

from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import ProblemDefinition, SolverParameters

pytestmark = pytest.mark.gpu
EPS = np.finfo(float).eps
SQRT_EPS = np.sqrt(EPS)

Q11 = np.array([0.0, 0.0950285641187840757204697, 0.2659896360172592788551071, 0.4941147113506765831125733,
                0.7608818755930255584019051, 1.0479359055822168311777887, 1.3370090901260500704239575,
                1.6098424281109515732168802, 1.8481068641834854648919872, 2.0333242222438583368671061,
                2.1467874956452459578315484]).reshape(-1, 1)  # TO_test.cc:887-897


def compare(a, b, tol):  # CompareMatrices(relative), utils/eigen_matrix_compare.h:95-98
    a, b = np.asarray(a, float), np.asarray(b, float)
    return bool(np.all(np.abs(a - b) <= tol * np.maximum(1.0, np.maximum(np.abs(a), np.abs(b)))))


def pendulum(gravity=True):
    m = load_model("pendulum")
    if not gravity:
        m.gravity = np.zeros(3)
    return m


def mk(model, N, dt, q_init, v_init, Qq, Qv, Qfq, Qfv, R, q_nom, v_nom):
    nq, nv = model.nq, model.nv
    prob = ProblemDefinition(num_steps=N, q_init=np.atleast_1d(q_init).astype(float),
                             v_init=np.atleast_1d(v_init).astype(float), Qq=Qq * np.eye(nq), Qv=Qv * np.eye(nv),
                             Qf_q=Qfq * np.eye(nq), Qf_v=Qfv * np.eye(nv), R=R * np.eye(nv),
                             q_nom=np.tile(np.atleast_1d(q_nom).astype(float), (N + 1, 1)),
                             v_nom=np.tile(np.atleast_1d(v_nom).astype(float), (N + 1, 1)), time_step=dt)
    return hip.HipPath(model, prob, SolverParameters(verbose=False))


def test_pendulum_calc_inverse_dynamics():  # TO_test.cc:1314-1386
    N, dt = 5, 1e-2
    dev = mk(pendulum(), N, dt, 0.0, -0.23, 1, 1, 1, 1, 1, 0.0, 0.0)
    q = np.array([-0.2 + dt * 0.1 * t * t for t in range(N + 1)]).reshape(-1, 1)
    dev.set_q(q)
    dev.eval_tau()
    v, tau = dev.get("v"), dev.get("tau")
    m, l, b, g = 1.0, 0.5, 0.1, 9.81
    for t in range(N):
        acc = (v[t + 1, 0] - v[t, 0]) / dt
        tau_gt = m * l * l * acc + m * g * l * np.sin(q[t + 1, 0]) + b * v[t + 1, 0]
        assert compare(tau[t, 0], tau_gt, 4 * EPS)


def test_pendulum_dtau_dq():  # TO_test.cc:1058-1150
    N, dt = 5, 1e-2
    dev = mk(pendulum(), N, dt, 0.0, 0.1, 1, 1, 1, 1, 1, 0.0, 0.0)
    q = np.array([0.0] + [0.6 * t for t in range(1, N + 1)]).reshape(-1, 1)
    dev.set_q(q)
    dev.eval_partials()
    m, l, b, g = 1.0, 0.5, 0.1, 9.81
    P = {k: dev.get(k) for k in ("dtau_dqm", "dtau_dqt", "dtau_dqp")}
    for t in range(1, N):
        assert compare(P["dtau_dqp"][t], m * l * l / dt / dt + b / dt + m * g * l * np.cos(q[t + 1, 0]), SQRT_EPS)
        assert compare(P["dtau_dqt"][t], -2 * m * l * l / dt / dt - b / dt, SQRT_EPS)
        assert compare(P["dtau_dqm"][t], 0.0 if t == 1 else m * l * l / dt / dt, SQRT_EPS)
    assert np.isnan(P["dtau_dqm"][0, 0, 0]) and P["dtau_dqt"][0, 0, 0] == 0.0  # inverse_dynamics_partials.h:35-42


def test_calc_gradient_pendulum_no_gravity():  # TO_test.cc:848-998
    N, dt = 10, 5e-2
    dev = mk(pendulum(False), N, dt, 0.0, 0.0, 0.1, 0.2, 0.3, 0.4, 0.5, np.pi, -0.1)
    m, l, b = 1.0, 0.5, 0.1

    def cost(q):
        dev.set_q(q)
        dev.eval_tau()
        return dev.get("cost")

    dev.set_q(Q11)
    dev.eval_partials()
    v, a, tau = dev.get("v"), dev.get("a"), dev.get("tau")
    for t in range(N):
        assert abs(tau[t, 0] - (m * l * l * a[t, 0] + b * v[t + 1, 0])) <= 10 * EPS   # :995-1003
    for t in range(1, N):  # :975-982
        assert abs(dev.get("dtau_dqp")[t, 0, 0] - (m * l * l / dt / dt + b / dt)) < 10 * SQRT_EPS
        assert abs(dev.get("dtau_dqt")[t, 0, 0] - (-2 * m * l * l / dt / dt - b / dt)) < 10 * SQRT_EPS
    dev.grad_hess()
    g = dev.get("gradient")
    g_fd = np.zeros(N + 1)   # central difference of the device cost stands in for autodiff (:923-933)
    for t in range(1, N + 1):
        h = 1e-6
        qp, qm = Q11.copy(), Q11.copy()
        qp[t] += h
        qm[t] -= h
        g_fd[t] = (cost(qp) - cost(qm)) / (2 * h)
    assert compare(g, g_fd, 1e-6) and g[0] == 0.0


def test_calc_cost_from_state():  # TO_test.cc:1155-1246
    N, dt = 10, 5e-2
    dev = mk(pendulum(False), N, dt, 0.0, 0.0, 0.0, 0.1, 10.0, 1.0, 1.0, np.pi, -0.1)
    dev.set_q(Q11)
    dev.eval_tau()
    L = dev.get("cost")
    m, l, b = 1.0, 0.5, 0.1
    q = Q11[:, 0]
    L_gt, vt = 0.0, 0.0
    for t in range(N):
        if t > 0:
            vt = (q[t] - q[t - 1]) / dt
        vp = (q[t + 1] - q[t]) / dt
        ut = m * l * l * (vp - vt) / dt + b * vp
        L_gt += dt * (vt + 0.1) * 0.1 * (vt + 0.1) + dt * ut * 1.0 * ut
    vt = (q[N] - q[N - 1]) / dt
    L_gt += (q[N] - np.pi) * 10.0 * (q[N] - np.pi) + (vt + 0.1) * 1.0 * (vt + 0.1)
    assert abs(L - L_gt) <= 100 * EPS * max(1.0, abs(L_gt))


def test_calc_velocities():  # TO_test.cc:1394-1443
    N, dt = 5, 1e-2
    v_init = np.array([0.5 / dt, 1.5 / dt])
    dev = mk(load_model("acrobot"), N, dt, [0.1, 0.2], v_init, 1, 1, 1, 1, 1, [0, 0], [0, 0])
    q = np.array([[0.1 + 0.5 * t, 0.2 + 1.5 * t] for t in range(N + 1)])
    dev.set_q(q)
    dev.eval_tau()
    v = dev.get("v")
    for t in range(N + 1):
        assert compare(v[t], v_init, EPS / dt)
    assert np.array_equal(dev.get("nplus")[2], np.eye(2))


def test_hessian_is_gauss_newton_of_residuals():  # TO_test.cc:496-637 (HessianAcrobot)
    """H = J^T J for the weighted residual r(q); J by central differences of the device's own
    v and tau (the reference uses autodiff)."""
    N, dt = 6, 1e-2
    model = load_model("acrobot")
    dev = mk(model, N, dt, [0.2, 0.1], [0.0, 0.0], 0.1, 0.2, 0.3, 0.4, 0.5, [1.2, 1.1], [-1.1, 1.0])
    rng = np.random.default_rng(3)
    q = np.array([0.2, 0.1]) + 0.05 * rng.normal(size=(N + 1, 2)).cumsum(axis=0)
    q[0] = [0.2, 0.1]

    def resid(qq):
        dev.set_q(qq)
        dev.eval_tau()
        v, tau = dev.get("v"), dev.get("tau")
        r = []
        for t in range(N):
            r += list(np.sqrt(2 * dt * 0.1) * (qq[t] - [1.2, 1.1])) + list(np.sqrt(2 * dt * 0.2) * (v[t] - [-1.1, 1.0]))
            r += list(np.sqrt(2 * dt * 0.5) * tau[t])
        r += list(np.sqrt(2 * 0.3) * (qq[N] - [1.2, 1.1])) + list(np.sqrt(2 * 0.4) * (v[N] - [-1.1, 1.0]))
        return np.array(r)

    nvar = (N + 1) * 2
    J = np.zeros((len(resid(q)), nvar))
    for j in range(2, nvar):
        h = 1e-6
        qp, qm = q.copy(), q.copy()
        qp.flat[j] += h
        qm.flat[j] -= h
        J[:, j] = (resid(qp) - resid(qm)) / (2 * h)
    H_gn = J.T @ J
    dev.set_q(q)
    dev.eval_partials()
    dev.grad_hess()
    A, B, Cc = dev.get("H_A"), dev.get("H_B"), dev.get("H_C")
    H = np.zeros((nvar, nvar))
    for i in range(N + 1):
        H[2 * i:2 * i + 2, 2 * i:2 * i + 2] = Cc[i]
        if i >= 1:
            H[2 * i:2 * i + 2, 2 * i - 2:2 * i] = B[i]
            H[2 * i - 2:2 * i, 2 * i:2 * i + 2] = B[i].T
        if i >= 2:
            H[2 * i:2 * i + 2, 2 * i - 4:2 * i - 2] = A[i]
            H[2 * i - 4:2 * i - 2, 2 * i:2 * i + 2] = A[i].T
    assert np.abs(H[2:, 2:] - H_gn[2:, 2:]).max() <= 1e-5 * np.abs(H).max()
    assert np.array_equal(H[:2, :2], np.eye(2)) and not H[:2, 2:].any()     # q_0 decoupled (TO.cc:1110-1113)


def test_contact_gradient_methods():  # TO_test.cc:183-280 (the autodiff leg replaced by central differences)
    """spinner_sphere, dt = 1, N = 2, the reference's q: tau is the same whatever the gradient
    method, forward / central / 4th-order central partials agree to the reference's tolerances,
    and each equals the oracle's bit for bit"""
    from idto_amd.problem import ProblemDefinition, SolverParameters
    from oracle_lib import Oracle
    model = load_model("spinner_sphere")
    N, dt = 2, 1.0
    q = np.array([[0.2, 1.5, 0.0], [0.4, 1.5, 0.0], [0.3, 1.4, 0.0]])
    sq = np.sqrt(np.finfo(float).eps)
    out = {}
    for meth in ("forward_differences", "central_differences", "central_differences4"):
        prob = ProblemDefinition(num_steps=N, q_init=q[0], v_init=np.zeros(3), Qq=np.eye(3), Qv=np.eye(3),
                                 Qf_q=np.eye(3), Qf_v=np.eye(3), R=np.eye(3), q_nom=np.zeros((N + 1, 3)),
                                 v_nom=np.zeros((N + 1, 3)), time_step=dt)
        sp = SolverParameters(verbose=False, gradients_method=meth)
        dev = hip.HipPath(model, prob, sp)
        dev.set_q(q)
        dev.eval_partials()
        out[meth] = (dev.get("tau"), {k: dev.get(k) for k in ("dtau_dqm", "dtau_dqt", "dtau_dqp")})
        P = Oracle(model, prob, sp).eval_partials(q)
        for k in ("dtau_dqm", "dtau_dqt", "dtau_dqp"):
            a, b = out[meth][1][k], np.asarray(P[k])
            assert np.all((a == b) | (np.isnan(a) & np.isnan(b))), (meth, k)
        dev.close()
    tau_f, Pf = out["forward_differences"]
    tau_c, Pc = out["central_differences"]
    _, Pc4 = out["central_differences4"]
    assert np.array_equal(tau_f, tau_c)
    assert np.abs(tau_f).max() > 0.1  # the contact is active in this configuration
    close = lambda x, y, tol: bool(np.all(np.abs(x - y) <= tol * np.maximum(1.0, np.maximum(np.abs(x), np.abs(y)))))
    for k in ("dtau_dqm", "dtau_dqt", "dtau_dqp"):
        for t in range(1, N):
            assert close(Pf[k][t], Pc[k][t], 100 * sq)
            assert close(Pc4[k][t], Pc[k][t], 100 * sq)
