"""ProblemDefinition / SolverParameters mirrors and the example-config loader.

Mirrors (same field names, meaning and defaults):
  * ``ProblemDefinition``  — reference optimizer/problem_definition.h:24-59
  * ``SolverParameters``   — reference optimizer/solver_parameters.h:64-167
  * ``make_problem``       — reference examples/example_base.cc:377-426
    (``SetProblemDefinition``) and :428-543 (``SetSolverParameters``)
  * ``make_linear_interpolation`` — reference examples/example_base.h:195-205
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field

import numpy as np
import yaml

from .model import CContactParams, CProblem, CSolverParams, Model, dptr, load_model

CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")

# enum values in the reference's declaration order (solver_parameters.h:14-62)
LINESEARCH = {"armijo": 0, "backtracking": 1}
METHOD = {"linesearch": 0, "trust_region": 1}
GRADIENTS = {"forward_differences": 0, "central_differences": 1, "central_differences4": 2, "autodiff": 3,
             "no_gradients": 4}
SCALING = {"sqrt": 0, "adaptive_sqrt": 1, "double_sqrt": 2, "adaptive_double_sqrt": 3}
LINEAR_SOLVER = {"dense_ldlt": 0, "pentadiagonal_lu": 1}


@dataclass
class ProblemDefinition:
    num_steps: int
    q_init: np.ndarray
    v_init: np.ndarray
    Qq: np.ndarray
    Qv: np.ndarray
    Qf_q: np.ndarray
    Qf_v: np.ndarray
    R: np.ndarray
    q_nom: np.ndarray  # (N+1, nq)
    v_nom: np.ndarray  # (N+1, nv)
    time_step: float = 0.05  # the reference reads this from the plant (MultibodyPlantConfig.time_step)

    def to_c(self):
        f = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64))
        # dense weights are symmetric in every use; stored column-major like Eigen
        keep = dict(q_init=f(self.q_init), v_init=f(self.v_init), Qq=f(np.asarray(self.Qq).T),
                    Qv=f(np.asarray(self.Qv).T), Qf_q=f(np.asarray(self.Qf_q).T), Qf_v=f(np.asarray(self.Qf_v).T),
                    R=f(np.asarray(self.R).T), q_nom=f(self.q_nom), v_nom=f(self.v_nom))
        p = CProblem()
        p.num_steps, p.time_step = int(self.num_steps), float(self.time_step)
        for k, v in keep.items():
            setattr(p, k, dptr(v))
        return p, keep


@dataclass
class SolverParameters:
    check_convergence: bool = False
    rel_cost_reduction: float = 0.0
    abs_cost_reduction: float = 0.0
    rel_gradient_along_dq: float = 0.0
    abs_gradient_along_dq: float = 0.0
    rel_state_change: float = 0.0
    abs_state_change: float = 0.0
    method: str = "trust_region"
    linesearch_method: str = "armijo"
    max_iterations: int = 100
    max_linesearch_iterations: int = 50
    gradients_method: str = "forward_differences"
    linear_solver: str = "pentadiagonal_lu"
    normalize_quaternions: bool = False
    verbose: bool = True
    print_debug_data: bool = False
    debug_compare_against_dense: bool = False
    linesearch_plot_every_iteration: bool = False
    save_contour_data: bool = False
    save_lineplot_data: bool = False
    contact_stiffness: float = 100.0
    dissipation_velocity: float = 0.1
    stiction_velocity: float = 0.05
    friction_coefficient: float = 0.5
    smoothing_factor: float = 0.1
    exact_hessian: bool = False
    scaling: bool = True
    scaling_method: str = "double_sqrt"
    equality_constraints: bool = True
    Delta0: float = 1e-1
    Delta_max: float = 1e5
    num_threads: int = 1
    q_nom_relative_to_q_init: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=bool))

    def to_c(self):
        s = CSolverParams()
        s.check_convergence = int(self.check_convergence)
        for k in ("rel_cost_reduction", "abs_cost_reduction", "rel_gradient_along_dq", "abs_gradient_along_dq",
                  "rel_state_change", "abs_state_change", "Delta0", "Delta_max"):
            setattr(s, k, float(getattr(self, k)))
        s.method = METHOD[self.method]
        s.linesearch_method = LINESEARCH[self.linesearch_method]
        s.max_iterations = int(self.max_iterations)
        s.max_linesearch_iterations = int(self.max_linesearch_iterations)
        s.gradients_method = GRADIENTS[self.gradients_method]
        s.linear_solver = LINEAR_SOLVER[self.linear_solver]
        s.normalize_quaternions = int(self.normalize_quaternions)
        s.verbose = int(self.verbose)
        s.scaling = int(self.scaling)
        s.scaling_method = SCALING[self.scaling_method]
        s.equality_constraints = int(self.equality_constraints)
        s.num_threads = int(self.num_threads)
        s.print_debug_data = int(self.print_debug_data)
        s.debug_compare_against_dense = int(self.debug_compare_against_dense)
        s.exact_hessian = int(self.exact_hessian)
        s.plot_dumps = int(self.linesearch_plot_every_iteration or self.save_contour_data or self.save_lineplot_data)
        return s

    def contact_to_c(self):
        return CContactParams(float(self.contact_stiffness), float(self.dissipation_velocity),
                              float(self.stiction_velocity), float(self.friction_coefficient),
                              float(self.smoothing_factor))


def load_config(name: str) -> dict:
    path = name if os.path.exists(name) else os.path.join(CONFIG_DIR, name + ".yaml")
    return yaml.safe_load(open(path))


def make_linear_interpolation(start, end, n):
    """reference examples/example_base.h:195-205"""
    start, end = np.asarray(start, float), np.asarray(end, float)
    out = np.zeros((n, len(start)))
    for i in range(n):
        lam = i / (n - 1.0)
        out[i] = (1 - lam) * start + lam * end
    return out


def normalize_quaternions(model: Model, q):
    q = np.array(q, dtype=float)
    for qs in model.quaternion_starts:
        q[..., qs:qs + 4] /= np.linalg.norm(q[..., qs:qs + 4], axis=-1, keepdims=True)
    return q


def make_problem(cfg: dict, model: Model | None = None, num_steps: int | None = None):
    """Returns (ProblemDefinition, SolverParameters, q_guess[(N+1), nq]) built the way
    ``TrajOptExample::SolveTrajectoryOptimization`` does (example_base.cc:189-248)."""
    if model is None:
        model = load_model(cfg["model"])
    N = int(cfg["num_steps"] if num_steps is None else num_steps)
    dt = float(cfg["time_step"])
    q_init = np.asarray(cfg["q_init"], float)
    v_init = np.asarray(cfg["v_init"], float)
    rel = np.asarray(cfg.get("q_nom_relative_to_q_init", [False] * len(q_init)), dtype=bool)
    q_nom_start = np.asarray(cfg["q_nom_start"], float) + rel * q_init
    q_nom_end = np.asarray(cfg["q_nom_end"], float) + rel * q_init
    q_nom = make_linear_interpolation(q_nom_start, q_nom_end, N + 1)
    v_nom = np.zeros((N + 1, len(v_init)))
    v_nom[0] = v_init
    for t in range(1, N + 1):
        v_nom[t] = (q_nom[t] - q_nom[t - 1]) / dt if len(q_init) == len(v_init) else v_init
    q_nom = normalize_quaternions(model, q_nom)
    q_init = normalize_quaternions(model, q_init)
    prob = ProblemDefinition(num_steps=N, q_init=q_init, v_init=v_init,
                             Qq=np.diag(np.asarray(cfg["Qq"], float)), Qv=np.diag(np.asarray(cfg["Qv"], float)),
                             Qf_q=np.diag(np.asarray(cfg["Qfq"], float)), Qf_v=np.diag(np.asarray(cfg["Qfv"], float)),
                             R=np.diag(np.asarray(cfg["R"], float)), q_nom=q_nom, v_nom=v_nom, time_step=dt)
    sp = SolverParameters()
    sp.max_iterations = int(cfg.get("max_iters", sp.max_iterations))
    sp.method = cfg.get("method", sp.method)
    sp.linesearch_method = cfg.get("linesearch", sp.linesearch_method)
    sp.max_linesearch_iterations = 60  # example_base.cc:474 hard-codes 60
    sp.gradients_method = cfg.get("gradients_method", sp.gradients_method)
    sp.linear_solver = cfg.get("linear_solver", sp.linear_solver)
    sp.normalize_quaternions = bool(cfg.get("normalize_quaternions", False))
    sp.scaling = bool(cfg.get("scaling", sp.scaling))
    sp.scaling_method = cfg.get("scaling_method", sp.scaling_method)
    sp.equality_constraints = bool(cfg.get("equality_constraints", sp.equality_constraints))
    sp.Delta0 = float(cfg.get("Delta0", sp.Delta0))
    sp.Delta_max = float(cfg.get("Delta_max", sp.Delta_max))
    sp.num_threads = int(cfg.get("num_threads", 1))
    for k in ("print_debug_data", "linesearch_plot_every_iteration", "save_contour_data", "save_lineplot_data",
              "exact_hessian"):   # examples/yaml_config.h:141-160, example_base.cc:452-493
        setattr(sp, k, bool(cfg.get(k, getattr(sp, k))))
    for k in ("contact_stiffness", "dissipation_velocity", "smoothing_factor", "friction_coefficient",
              "stiction_velocity"):
        if k in cfg:
            setattr(sp, k, float(cfg[k]))
    tol = cfg.get("tolerances") or {}
    for k, v in tol.items():
        setattr(sp, k, float(v))
    sp.q_nom_relative_to_q_init = rel
    sp.verbose = False
    q_guess = make_linear_interpolation(q_init, np.asarray(cfg["q_guess"], float), N + 1)
    q_guess = normalize_quaternions(model, q_guess)
    q_guess[0] = q_init
    return prob, sp, q_guess


# ---- synthetic trajectories for parity tests and the bench ----
def _splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return state, z ^ (z >> 31)


def synthetic_trajectory(cfg: dict, model: Model, num_steps: int, seed: int = 0, amplitude: float = 0.02,
                         lower: float = 0.0):
    """q_t = lerp(q_init, q_guess, t/N) + amplitude * U(-1,1), q_0 = q_init, quaternions
    re-normalised; `lower` drops the floating base / planar height so contacts are active."""
    q_init = np.asarray(cfg["q_init"], float)
    q = make_linear_interpolation(q_init, np.asarray(cfg["q_guess"], float), num_steps + 1)
    st = (seed * 0x632BE59BD9B4E019 + 1) & 0xFFFFFFFFFFFFFFFF
    for t in range(1, num_steps + 1):
        for i in range(len(q_init)):
            st, r = _splitmix64(st)
            u = (r >> 11) * (1.0 / (1 << 53))
            q[t, i] += amplitude * (2.0 * u - 1.0)
    if lower:
        for b in range(model.nbodies):
            jt, qs = int(model.jtype[b]), int(model.qstart[b])
            if jt == 3:
                q[1:, qs + 6] -= lower
            elif jt == 2 and model.parent[b] == -1:
                q[1:, qs] -= lower
    q = normalize_quaternions(model, q)
    q[0] = normalize_quaternions(model, q_init)
    return q
