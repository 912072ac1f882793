#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/mpc_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_mpc_timeline_${1:-x}.txt
timeout 200 python tools/mpc_latency.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_mpc_latency_${1:-x}.txt
