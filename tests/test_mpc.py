"""The MPC shell (idto_amd/mpc.py, reference examples/mpc_controller.cc): host logic on CPU with
a recording stand-in for the optimizer; closed loop on the GPU with the real one."""
import numpy as np
import pytest

from idto_amd.mpc import Interpolator, ModelPredictiveController
from idto_amd.problem import ProblemDefinition, SolverParameters


class _Sol:
    def __init__(self, q, v, tau):
        self.q, self.v, self.tau = q, v, tau


class _FakeOptimizer:
    """records the calls the controller makes; SolveFromWarmStart returns the guess unchanged"""

    def __init__(self, N, nq, dt):
        self._prob = ProblemDefinition(num_steps=N, q_init=np.zeros(nq), v_init=np.zeros(nq), Qq=np.eye(nq), Qv=np.eye(nq),
                                       Qf_q=np.eye(nq), Qf_v=np.eye(nq), R=np.eye(nq),
                                       q_nom=np.tile(np.arange(nq, dtype=float), (N + 1, 1)),
                                       v_nom=np.zeros((N + 1, nq)), time_step=dt)
        self._params = SolverParameters(q_nom_relative_to_q_init=np.array([True] + [False] * (nq - 1)))
        self.calls = []
        self.q = None

    def time_step(self): return self._prob.time_step
    def num_steps(self): return self._prob.num_steps
    def prob(self): return self._prob
    def params(self): return self._params

    def CreateWarmStart(self, q):
        outer = self

        class WS:
            def set_q(self, q): outer.q = np.array(q)
        outer.q = np.array(q)
        return WS()

    def UpdateNominalTrajectory(self, q_nom, v_nom):
        self.calls.append(("nom", np.array(q_nom)))
        self._prob.q_nom = np.array(q_nom)

    def ResetInitialConditions(self, q0, v0): self.calls.append(("init", np.array(q0), np.array(v0)))

    def SolveFromWarmStart(self, ws, sol, stats):
        N, nq = self.num_steps(), self.q.shape[1]
        sol.q, sol.v, sol.tau = self.q.copy(), np.zeros((N + 1, nq)), np.ones((N, nq))
        return "kMaxIterationsReached"


def test_initial_guess_is_the_time_shifted_previous_solution():
    N, nq, dt = 10, 2, 0.1
    opt = _FakeOptimizer(N, nq, dt)
    ts = dt * np.arange(N + 1)
    q = np.stack([np.sin(ts), ts ** 2], axis=1)
    mpc = ModelPredictiveController(opt, _Sol(q, np.zeros((N + 1, nq)), np.ones((N, nq))))
    # replan 0.25 s later from a measured state
    q0 = np.array([0.3, 0.07])
    traj = mpc.update(0.25, q0, np.array([0.1, 0.2]))
    guess = opt.q
    assert np.array_equal(guess[0], q0)                                   # mpc_controller.cc:57
    want = np.stack([np.sin(0.25 + ts), (0.25 + ts) ** 2], axis=1)         # UpdateInitialGuess :87-97
    assert np.abs(guess[1:8] - want[1:8]).max() < 5e-4                     # cubic interpolation / extrapolation
    # nominal trajectory shifted only for the selected DoF (:62-69)
    nom = [c for c in opt.calls if c[0] == "nom"][-1][1]
    assert np.allclose(nom[:, 0], 0.0 + (q0[0] - 0.0)) and np.allclose(nom[:, 1], 1.0)
    init = [c for c in opt.calls if c[0] == "init"][-1]
    assert np.array_equal(init[1], q0) and np.array_equal(init[2], [0.1, 0.2])
    # the stored trajectory restarts at the replan time; the interpolator reads it back
    assert traj.start_time == 0.25
    assert np.allclose(Interpolator.state(traj, 0.25)[:nq], q0)
    assert np.allclose(Interpolator.control(traj, 0.3), 1.0)


def test_selector_size_is_checked():
    opt = _FakeOptimizer(5, 2, 0.1)
    opt._params.q_nom_relative_to_q_init = np.zeros(0, dtype=bool)
    q = np.zeros((6, 2))
    mpc = ModelPredictiveController(opt, _Sol(q, q, np.zeros((5, 2))))
    with pytest.raises(ValueError):
        mpc.update(0.1, np.zeros(2), np.zeros(2))


@pytest.mark.gpu
def test_closed_loop_replanning_on_the_device():
    """spinner example (reference examples/spinner/spinner.yaml: mpc_iters 1, 200 Hz): open-loop
    solve, then replans from the predicted state - every replan must leave a finite cost and honour the
    measured initial condition."""
    from idto_amd.model import load_model
    from idto_amd.optimizer import TrajectoryOptimizer, TrajectoryOptimizerSolution, TrajectoryOptimizerStats
    from idto_amd.problem import load_config, make_problem
    cfg = load_config("spinner")
    model = load_model("spinner")
    prob, sp, q_guess = make_problem(cfg, model)
    sp.verbose, sp.max_iterations = False, 30
    sp.q_nom_relative_to_q_init = np.zeros(model.nq, dtype=bool)
    opt = TrajectoryOptimizer(model, prob, sp)
    sol, st = TrajectoryOptimizerSolution(), TrajectoryOptimizerStats()
    opt.Solve(q_guess, sol, st)
    sp1 = SolverParameters(**{**sp.__dict__, "max_iterations": int(cfg.get("mpc_iters", 1))})
    opt1 = TrajectoryOptimizer(model, prob, sp1)
    mpc = ModelPredictiveController(opt1, sol, actuated=model.actuated)
    period = 1.0 / float(cfg.get("controller_frequency", 200.0))
    costs = []
    for k in range(1, 4):
        t = k * period
        x = Interpolator.state(mpc.stored, t)
        traj = mpc.update(t, x[:model.nq], x[model.nq:])
        costs.append(mpc.last_stats.iteration_costs[0])
        assert np.allclose(traj.q(0.0), x[:model.nq], atol=1e-12)
        assert np.all(np.isfinite(Interpolator.control(traj, t + 0.5 * period)))
    # (the guess of a replan is the previous solution shifted in time and HELD at its last knot - PiecewisePolynomial::value
    # clamps - so its cost need not fall from replan to replan; it must stay finite and of the same order)
    assert np.all(np.isfinite(costs)) and max(costs) <= 10 * costs[0]


# ---- the C++ shell (include/idto/examples/mpc_controller.h, libidto_opt.so idto_mpc_*)

def test_cpp_spline_is_the_not_a_knot_cubic_clamped_to_its_range():
    """PiecewiseCubic == scipy's not-a-knot CubicSpline (the interpolant Drake's CubicWithContinuousSecondDerivatives
    builds, reference examples/mpc_controller.cc:130-137) inside the knots' range; outside, the value at the nearest
    end (drake PiecewisePolynomial::value clamps).  Uniform and non-uniform breaks, 2 / 3 / 4 / 41 knots."""
    from scipy.interpolate import CubicSpline
    from idto_amd.mpc import spline_eval
    rng = np.random.default_rng(7)
    for n, uniform in ((41, True), (41, False), (4, True), (5, False), (3, False), (2, True)):
        t = 0.05 * np.arange(n) if uniform else np.cumsum(rng.uniform(0.02, 0.2, n))
        y = np.stack([np.sin(3 * t) + 0.1 * rng.standard_normal(n), t ** 3 - t, rng.standard_normal(n)], axis=1)
        times = np.concatenate([t, np.linspace(t[0], t[-1], 201), [t[0] - 0.3, t[-1] + 0.7]])
        got = spline_eval(t, y, times)
        if n >= 4:
            ref = CubicSpline(t, y, bc_type="not-a-knot")
        elif n == 3:
            ref = lambda x: np.stack([np.polyval(np.polyfit(t, y[:, c], 2), x) for c in range(3)], axis=1)
        else:
            ref = lambda x: np.stack([np.interp(x, t, y[:, c]) for c in range(3)], axis=1)
        want = ref(np.clip(times, t[0], t[-1]))
        assert np.abs(got - want).max() <= 1e-11 * max(1.0, np.abs(y).max()), (n, uniform, np.abs(got - want).max())
        assert np.array_equal(got[-1], got[n - 1]) and np.array_equal(got[-2], got[0])   # clamped, not extrapolated


@pytest.mark.gpu
def test_cpp_shell_replans_like_the_python_shell():
    """Closed loop on the device with the C++ ModelPredictiveController against the numpy / scipy shell over the same
    optimizer settings (spinner: mpc_iters 1, 200 Hz; hopper: q_nom relative to q_init for the base x): the same
    initial guesses, solutions, interpolated states and controls, replan after replan."""
    from idto_amd.model import load_model
    from idto_amd.mpc import DeviceModelPredictiveController
    from idto_amd.optimizer import TrajectoryOptimizer, TrajectoryOptimizerSolution, TrajectoryOptimizerStats
    from idto_amd.problem import load_config, make_problem
    for name, sel_idx in (("spinner", []), ("hopper", [0])):
        cfg, model = load_config(name), load_model(name)
        prob, sp, q_guess = make_problem(cfg, model)
        sp.verbose, sp.max_iterations = False, 20
        sel = np.zeros(model.nq, dtype=bool)
        sel[sel_idx] = True
        sp.q_nom_relative_to_q_init = sel
        opt = TrajectoryOptimizer(model, prob, sp)
        sol, st = TrajectoryOptimizerSolution(), TrajectoryOptimizerStats()
        opt.Solve(q_guess, sol, st)
        sp1 = SolverParameters(**{**sp.__dict__, "max_iterations": int(cfg.get("mpc_iters", 1))})
        period = 1.0 / float(cfg.get("controller_frequency", 200.0))
        opt_py, opt_cc = TrajectoryOptimizer(model, prob, sp1), TrajectoryOptimizer(model, prob, sp1)
        py = ModelPredictiveController(opt_py, sol, actuated=model.actuated)
        cc = DeviceModelPredictiveController(opt_cc, sol, actuated=model.actuated, q_nom_relative_to_q_init=sel, replan_period=period)
        assert cc.nu == int(np.asarray(model.actuated, bool).sum() or model.nv)
        scale = max(1.0, np.abs(np.asarray(sol.q)).max())
        for k in range(1, 5):
            t = k * period
            x_py, x_cc = Interpolator.state(py.stored, t), cc.state(t)
            assert np.abs(x_py - x_cc).max() <= 1e-10 * scale
            x = x_cc + (0.01 if k == 2 else 0.0)          # a disturbed state estimate at the second replan
            q0, v0 = x[:model.nq], x[model.nq:]
            traj = py.update(t, q0, v0)
            g, q, v, tau = cc.update(t, q0, v0)
            assert np.array_equal(g[0], q0) and cc.start_time == t
            assert all(np.array_equal(a, b) and a is not b for a, b in zip((g, q, v, tau), cc._buf[1:5]))   # (copies of the controller's buffers; copy=False hands out the buffers)
            assert cc.last_flag in (0, 3)   # (kSuccess or kMaxIterationsReached: the example runs mpc_iters iterations)
            assert np.abs(g - opt_py_guess(py, opt_py)).max() <= 1e-9 * scale
            assert np.abs(q - np.asarray(traj.q(py.time_step * np.arange(py.num_steps)))).max() <= 1e-7 * scale
            for tq in (t, t + 0.4 * period, t + 3.3 * period, t + 100.0):
                assert np.abs(cc.state(tq) - Interpolator.state(traj, tq)).max() <= 1e-7 * scale
                assert np.abs(cc.control(tq) - Interpolator.control(traj, tq)).max() <= 1e-6 * max(1.0, np.abs(tau).max())
            assert np.isfinite(cc.last_cost)
        cc.close(); opt_py.close(); opt_cc.close(); opt.close()


def opt_py_guess(py, opt_py):
    """the initial guess the Python shell handed to its optimizer in the last update"""
    return py._last_guess
