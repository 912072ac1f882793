#!/usr/bin/env python3
"""Build-time guard on what the register allocator did to the production kernels (VERDICT r4 "weak" #1 iv).

DESIGN.md section 10: an inlined tail once pushed fd_kernel's SGPR spills from 105 to 146 and the kernel then computed
wrong partials without the tail ever running - the hot kernels sit at the edge of what the allocator handles, and only
the parity tests on a GPU would have noticed.  build.sh compiles with -Rpass-analysis=kernel-resource-usage, keeps the
remarks under build/, and runs this script: a production instantiation with spilled vector registers, more scratch
or more spilled scalar registers than the limits below FAILS THE BUILD (exit 1).  tests/test_build_resources.py runs
the same check on the CPU (hipcc cross-compiles without a GPU).

usage: check_resources.py [--compile] [remarks files ...]     (--compile: produce the remarks first, ~90 s)
"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REMARKS = [os.path.join(ROOT, "build", "fd_launch.remarks"), os.path.join(ROOT, "build", "idto_hip.remarks")]

# kernel (demangled, without "idto_dev::" and arguments) -> (max spilled VGPRs, max scratch bytes per lane, max spilled SGPRs)
# Values: what profiles/r05_isa_resources.txt records, the SGPR figure with a margin of 24 (it moves by a few with every
# edit and is harmless up to there; 41 more than today's was the incident).
LIMITS = {
    "fd_kernel<2, 1>": (0, 20, 83 + 24),
    "fd_kernel<3, 2>": (0, 20, 87 + 24),
    "fd_kernel<3, 3>": (0, 20, 105 + 24),
    "fd_kernel<3, 5>": (0, 20, 85 + 24),
    "fd_kernel<4, 4>": (0, 20, 107 + 24),
    "penta_pipe_kernel<19, false>": (0, 36, 102 + 24),
    "penta_pipe_kernel<2, false>": (0, 0, 137 + 24),
    "penta_pipe_kernel<3, false>": (0, 0, 106 + 24),
    "penta_pipe_kernel<5, false>": (0, 0, 126 + 24),
    # (the instantiations that also decide on the trial point inside idto_hip_tr_solve: one more role, cost_kernel's work)
    "penta_pipe_kernel<19, true>": (0, 36, 204 + 24),
    "penta_pipe_kernel<5, true>": (0, 0, 209 + 24),
    # (round 6: the solver's workgroup became penta_band_body, shared with gn_small.h - the scalar registers of the inlined
    # body are allocated differently, 22 / 48 more of them spill to lanes; the step is unchanged: 26.9 / 38.0 us before and after)
    "penta_band_kernel<6>": (0, 112, 26 + 24),
    "penta_band_kernel<9>": (0, 112, 26 + 24),
    "penta_band_kernel<12>": (0, 112, 66 + 24),
    "penta_band_kernel<15>": (0, 112, 70 + 24),
    # (round 6, second half: the kernel also carries the trust-region loop's cost and decision - 130 / 137 scalar registers
    # spilled to lanes; the plain step's time is what it was, profiles/r06_all_configs.txt)
    "gn_small_kernel<1, 6, 256, 6, false>": (0, 152, 52 + 24),
    "gn_small_kernel<5, 9, 256, 9, false>": (0, 152, 56 + 24),
    # (the trust-region loop's instantiations: cost and decision behind the records, tr_iter_kernel's part in front - a whole
    # iteration of a small model in one launch -, with the enforced constraint the KKT system's blocks of nq + 1)
    "gn_small_kernel<1, 6, 256, 6, true>": (0, 152, 175 + 24),
    "gn_small_kernel<5, 9, 256, 9, true>": (0, 152, 175 + 24),
    "gn_small_kernel<1, 6, 256, 9, true>": (0, 152, 220 + 24),
    "gn_small_kernel<5, 9, 256, 12, true>": (0, 152, 228 + 24),
    "penta_nd_kernel<23, false>": (0, 0, 585 + 24),
    # ("a guard whose limit equals today's spill count guards nothing".  What it guards is the scratch column:
    # the six registers are spilled to ACCUMULATION registers - v_accvgpr_write / _read, no memory - which a kernel of one
    # wavefront per SIMD has 256 of; the limit that matters, 0 bytes of scratch, holds.  Getting the six back would need the
    # 29-row elimination's multipliers out of registers, i.e. penta_pipe.h's read-back scheme in penta_ldl.h: not done.)
    "penta_nd_kernel<29, false>": (6, 0, 869 + 24),
    "assemble_terms_kernel": (0, 0, 6 + 24),
    # (round 6: four instantiations of the row body - 4, 8, 20, 32 band entries per thread in registers - are inlined into
    # tr_iter_kernel, and the kernel carries the workgroup through the whole iteration: 29 -> 70 scalar registers spilled to lanes)
    # (68 B of private segment are RESERVED since the KKT solution is taken apart in here - a stack object whose accesses were
    # all promoted to registers: the kernel has no scratch instruction, build/isa/main.s)
    "tr_iter_kernel": (0, 68, 73 + 24),
    "cost_kernel": (0, 0, 17 + 24),
}


def compile_remarks():
    """The two translation units with the remark pass on (device code only, no object files kept)."""
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Iinclude", "-Iidto_amd/csrc", "-S",
             "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage"]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    jobs = [([hipcc] + flags + ["idto_amd/csrc/fd_launch.hip", "-o", "/dev/null"], REMARKS[0]),
            ([hipcc] + flags + ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "idto_amd/csrc/idto_hip.hip", "-o", "/dev/null"], REMARKS[1])]
    procs = [(subprocess.Popen(cmd, cwd=ROOT, stderr=open(out, "w")), out) for cmd, out in jobs]
    for p, out in procs:
        if p.wait() != 0:
            sys.stderr.write(open(out).read()[-4000:])
            raise SystemExit("check_resources: compiling for the remarks failed")


def parse(files):
    rows = {}
    for f in files:
        cur = None
        for line in open(f, errors="replace"):
            m = re.search(r"remark: (?:Function Name: (\S+)|\s*(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill): (\d+))", line)
            if not m:
                continue
            if m.group(1):
                cur = m.group(1); rows[cur] = {}
            elif cur:
                rows[cur][m.group(2)] = int(m.group(3))
    names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.split("\n")
    out = {}
    for mangled, name in zip(rows, names):
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(.*", "", name).replace("idto_dev::", "")
        out[name] = rows[mangled]
    return out


def check(files=None):
    """-> (list of violations, table of what was checked)"""
    res = parse(files or REMARKS)
    bad, table = [], []
    for k, (vs, sc, ss) in LIMITS.items():
        if k not in res:
            bad.append(f"{k}: not in the compiler's remarks (instantiation renamed or removed? update tools/check_resources.py)")
            continue
        d = res[k]
        got = (d.get("VGPRs Spill", 0), d.get("ScratchSize [bytes/lane]", 0), d.get("SGPRs Spill", 0))
        table.append((k, got, (vs, sc, ss)))
        if got[0] > vs: bad.append(f"{k}: {got[0]} spilled VGPRs (limit {vs})")
        if got[1] > sc: bad.append(f"{k}: {got[1]} B of scratch per lane (limit {sc})")
        if got[2] > ss: bad.append(f"{k}: {got[2]} spilled SGPRs (limit {ss})")
    return bad, table


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--compile"]
    if "--compile" in sys.argv[1:]:
        compile_remarks()
    bad, table = check(args or None)
    for k, got, lim in table:
        print(f"{k:32s} spilled VGPRs {got[0]:3d} (<= {lim[0]:3d})  scratch {got[1]:4d} B (<= {lim[1]:4d})  spilled SGPRs {got[2]:3d} (<= {lim[2]:3d})")
    if bad:
        print("RESOURCE CHECK FAILED:\n  " + "\n  ".join(bad))
        sys.exit(1)
    print("resource check passed")
