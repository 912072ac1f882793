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


def test_solver_parameter_defaults():
    """python_bindings/test/solver_parameters_test.py:8-58 (the defaults of solver_parameters.h:76-166),
    for the Python mirror and for the C++ header (compiled and printed)"""
    import os
    import subprocess
    import tempfile
    from idto_amd.problem import SolverParameters
    sp = SolverParameters()
    assert (sp.max_iterations, sp.max_linesearch_iterations, sp.num_threads) == (100, 50, 1)
    assert (sp.contact_stiffness, sp.dissipation_velocity, sp.stiction_velocity) == (100.0, 0.1, 0.05)
    assert (sp.friction_coefficient, sp.smoothing_factor) == (0.5, 0.1)
    assert sp.scaling is True and sp.equality_constraints is True and sp.verbose is True
    assert (sp.Delta0, sp.Delta_max) == (0.1, 1e5)
    assert sp.method == "trust_region" and sp.linesearch_method == "armijo"
    assert sp.gradients_method == "forward_differences" and sp.scaling_method == "double_sqrt"
    assert not sp.check_convergence and not sp.exact_hessian and not sp.normalize_quaternions
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = r'''
#include <cstdio>
#include "idto/optimizer/solver_parameters.h"
int main() {
  idto::optimizer::SolverParameters p;
  std::printf("%d %d %d %g %g %g %g %g %d %d %d %g %g %d %d %d %d %d\n", p.max_iterations, p.max_linesearch_iterations,
              p.num_threads, p.contact_stiffness, p.dissipation_velocity, p.stiction_velocity, p.friction_coefficient,
              p.smoothing_factor, (int)p.scaling, (int)p.equality_constraints, (int)p.verbose, p.Delta0, p.Delta_max,
              (int)p.method, (int)p.linesearch_method, (int)p.gradients_method, (int)p.scaling_method,
              (int)p.linear_solver);
}'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.cc"), "w").write(src)
        subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(root, "include"), os.path.join(d, "t.cc"),
                               "-o", os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    assert out == ["100", "50", "1", "100", "0.1", "0.05", "0.5", "0.1", "1", "1", "1", "0.1", "100000",
                   "1", "0", "0", "2", "1"], out
