"""CPU-side checks of the C-ABI boundary: libidto_hip.so loads without a GPU and
exports every symbol include/idto_hip.h declares; creating a context without a GPU
fails loudly instead of falling back to a CPU path."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "idto_hip.h")).read()
    return sorted(set(re.findall(r"\b(idto_hip_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported():
    if not os.path.exists(hip.LIB_PATH):
        pytest.fail(f"{hip.LIB_PATH} missing: run ./build.sh (the driver's build() does)")
    L = C.CDLL(hip.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for s in declared:
        assert hasattr(L, s), f"{s} declared in include/idto_hip.h but not exported"
    assert sorted(hip.EXPORTED_SYMBOLS) == declared


def test_optimizer_header_symbols_are_exported():
    """include/idto_opt.h (host-side TrajectoryOptimizer) <-> libidto_opt.so"""
    from idto_amd import optimizer
    if not os.path.exists(optimizer.LIB_PATH):
        pytest.fail(f"{optimizer.LIB_PATH} missing: run ./build.sh")
    text = open(os.path.join(ROOT, "include", "idto_opt.h")).read()
    declared = sorted(set(re.findall(r"\b(idto_opt_[a-z_0-9]+)\s*\(", text)))
    L = optimizer.lib()
    for s in declared:
        assert hasattr(L, s), f"{s} declared in include/idto_opt.h but not exported"
    assert sorted(optimizer.EXPORTED_SYMBOLS) == declared
    mpc = sorted(set(re.findall(r"\b(idto_mpc_[a-z_0-9]+)\s*\(", text)))   # the model-predictive-control shell
    for s in mpc:
        assert hasattr(L, s), f"{s} declared in include/idto_opt.h but not exported"
    assert sorted(optimizer.MPC_SYMBOLS) == mpc


def test_optimizer_has_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from idto_amd.optimizer import TrajectoryOptimizer
    cfg = load_config("spinner")
    model = load_model("spinner")
    prob, sp, _ = make_problem(cfg, model)
    with pytest.raises(RuntimeError, match="no HIP device"):
        TrajectoryOptimizer(model, prob, sp)


def test_host_api_headers_compile_standalone(tmp_path):
    """the C++ host API (include/idto/optimizer/*.h) is self-contained C++17"""
    import subprocess
    src = tmp_path / "t.cc"
    src.write_text('#include "idto/optimizer/trajectory_optimizer.h"\n'
                   'using namespace idto::optimizer;\n'
                   'int main() { SolverParameters p; ProblemDefinition d; TrajectoryOptimizerStats<double> s;\n'
                   '  return (p.max_iterations == 100 && p.scaling && p.equality_constraints && p.Delta0 == 0.1 &&\n'
                   '          p.method == kTrustRegion && p.scaling_method == kDoubleSqrt && s.is_empty() &&\n'
                   '          d.num_steps == 0) ? 0 : 1; }\n')
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src),
                           "-o", str(exe), "-L", os.path.join(ROOT, "idto_amd"), "-lidto_opt", "-lidto_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "idto_amd")])
    assert subprocess.call([str(exe)]) == 0


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    cfg = load_config("spinner")
    model = load_model("spinner")
    prob, sp, _ = make_problem(cfg, model)
    with pytest.raises(hip.HipError, match="no HIP device"):
        hip.HipPath(model, prob, sp)


def test_product_does_not_import_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may import, link or execute
    anything under oracle/ (comments may mention it)."""
    bad = re.compile(r'#include\s*[<"][^>"]*(oracle|rigid_body\.h|traj_opt\.h|penta\.h)|import\s+oracle|'
                     r'from\s+oracle|liboracle|oracle_lib|orc_[a-z_]+\(')
    for top in ("idto_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cc", ".cpp", ".sh")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    m = bad.search(text)
                    assert m is None, f"{os.path.join(dirpath, f)} uses the oracle: {m.group(0)}"
    assert "oracle" not in open(os.path.join(ROOT, "build.sh")).read()
