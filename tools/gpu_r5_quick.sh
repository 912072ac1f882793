#!/bin/bash
# Round 5 quick look: solver timelines (alone / with the assembly inside) and a short bench line.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python tools/nd_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5q_nd_timeline.txt
IDTO_TIMELINE_GN_STEP=1 timeout 120 python tools/nd_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5q_nd_timeline_gn.txt
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu 2>&1 | tail -1 | tee gpurun_out/r5q_bench.json
