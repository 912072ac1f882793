#!/bin/bash
# Registers, scratch, spills and occupancy of every kernel of libidto_hip.so as the compiler reports them
# (-Rpass-analysis=kernel-resource-usage; CPU only, hipcc cross-compiles): profiles/<round>_isa_resources.txt
# usage: tools/isa_resources.sh [round]
R=${1:-r04}
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Iidto_amd/csrc -S --cuda-device-only -Rpass-analysis=kernel-resource-usage"
mkdir -p build/isa
( /opt/rocm/bin/hipcc $FLAGS idto_amd/csrc/fd_launch.hip -o build/isa/fd.s 2> build/isa/fd.txt ) &
/opt/rocm/bin/hipcc $FLAGS -mllvm -amdgpu-mfma-vgpr-form=1 idto_amd/csrc/idto_hip.hip -o build/isa/main.s 2> build/isa/main.txt
wait
python3 - "$R" <<'PY'
import re, subprocess, sys
rows = {}
for f in ("build/isa/fd.txt", "build/isa/main.txt"):
    cur = None
    for line in open(f, errors="replace"):
        m = re.search(r"remark: (?:Function Name: (\S+)|\s*(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\d+))", line)
        if not m:
            continue
        if m.group(1):
            cur = m.group(1); rows[cur] = {}
        elif cur:
            rows[cur][m.group(2)] = int(m.group(3))
names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.split("\n")
out = []
for mangled, name in zip(rows, names):
    d = rows[mangled]
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*", "", name).replace("idto_dev::", "")
    out.append((name, d))
out.sort()
hdr = ("ISA resources of every kernel of libidto_hip.so (hipcc -O3 --offload-arch=gfx950 -ffp-contract=off,\n"
       "-Rpass-analysis=kernel-resource-usage; tools/isa_resources.sh: csrc/fd_launch.hip and csrc/idto_hip.hip).  Scratch = bytes per lane\n"
       "(20 B with 0 spilled VGPRs: SGPRs spilled to memory).  fd_kernel<MAXC, SHAPE>: SHAPE 0 = id_eval<MAXC> (any model),\n"
       "1 acrobot, 2 hopper, 3 mini_cheetah, 4 allegro_hand + ball, 5 spinner (id_fast.h).  Static LDS only.\n\n")
with open(f"profiles/{sys.argv[1]}_isa_resources.txt", "w") as fo:
    fo.write(hdr)
    fo.write(f"{'kernel':58s}{'VGPR':>6s}{'AGPR':>6s}{'SGPR':>6s}{'scratch':>9s}{'sgpr-spill':>11s}{'vgpr-spill':>11s}{'LDS':>8s}{'occ':>5s}\n")
    for name, d in out:
        fo.write(f"{name[:57]:58s}{d.get('VGPRs',0):6d}{d.get('AGPRs',0):6d}{d.get('SGPRs',0):6d}{d.get('ScratchSize [bytes/lane]',0):9d}"
                 f"{d.get('SGPRs Spill',0):11d}{d.get('VGPRs Spill',0):11d}{d.get('LDS Size [bytes/block]',0):8d}{d.get('Occupancy [waves/SIMD]',0):5d}\n")
print(open(f"profiles/{sys.argv[1]}_isa_resources.txt").read())
PY
