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
    solve, then replans from the predicted state - every replan must leave a finite, non-increasing
    cost and honour the measured initial condition."""
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
    assert np.all(np.isfinite(costs)) and costs[-1] <= costs[0] * 1.05
