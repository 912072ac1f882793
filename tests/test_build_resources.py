"""CPU-side guard on the register allocator (VERDICT r4 "weak" #1 iv, "next" #4a): the production instantiations of
fd_kernel and of the solvers must not spill vector registers, and their scratch / spilled-SGPR figures must stay at
what profiles/r05_isa_resources.txt records (+ a margin on the SGPRs).  DESIGN.md section 10 has the incident this
guards against: 105 -> 146 spilled SGPRs in fd_kernel and silently wrong partials.

build.sh writes the compiler's remarks (build/*.remarks) and fails by itself; this test re-reads them - or produces
them when they are missing or older than the sources (hipcc cross-compiles here, ~90 s) - so that the limits are also
enforced where only `pytest -m "not gpu"` runs.
"""
import glob
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_resources", os.path.join(ROOT, "tools", "check_resources.py"))
cr = importlib.util.module_from_spec(spec)
spec.loader.exec_module(cr)


def _stale():
    srcs = glob.glob(os.path.join(ROOT, "idto_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "idto_amd", "csrc", "*.hip")) + \
        glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(ROOT, "include", "idto", "*.h"))
    newest = max(os.path.getmtime(f) for f in srcs)
    return any(not os.path.exists(r) or os.path.getmtime(r) < newest or os.path.getsize(r) == 0 for r in cr.REMARKS)


def test_production_kernels_stay_within_their_register_budget():
    if _stale():
        cr.compile_remarks()
    bad, table = cr.check()
    assert len(table) >= len(cr.LIMITS) - len([b for b in bad if "not in the compiler's remarks" in b])
    assert not bad, "\n".join(bad)


def test_the_check_fails_on_a_kernel_over_its_limits(tmp_path):
    """the incident itself, as a remarks file: 146 spilled SGPRs in fd_kernel<3, 3> must be refused"""
    txt = open(cr.REMARKS[0], errors="replace").read() if os.path.exists(cr.REMARKS[0]) else ""
    fake = tmp_path / "fd.remarks"
    mangled = "_ZN8idto_dev9fd_kernelILi3ELi3EEEvNS_6FdArgsE"
    import re
    m = re.search(r"Function Name: (\S*fd_kernelILi3ELi3E\S*)", txt)
    if m:
        mangled = m.group(1)
    fake.write_text(f"x.h:1:1: remark: Function Name: {mangled} [-Rpass-analysis=kernel-resource-usage]\n"
                    "x.h:1:1: remark:     VGPRs: 256 [-Rpass-analysis=kernel-resource-usage]\n"
                    "x.h:1:1: remark:     ScratchSize [bytes/lane]: 20 [-Rpass-analysis=kernel-resource-usage]\n"
                    "x.h:1:1: remark:     SGPRs Spill: 146 [-Rpass-analysis=kernel-resource-usage]\n"
                    "x.h:1:1: remark:     VGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]\n")
    bad, _ = cr.check([str(fake)])
    assert any("fd_kernel<3, 3>" in b and "spilled SGPRs" in b for b in bad), bad
