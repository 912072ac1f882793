#!/bin/bash
# usage: gpu_r5_quicktests.sh "<pytest args>"  [then the MPC latency]
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest $1 -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -12
if [ -n "$2" ]; then timeout 200 python tools/mpc_latency.py 2>&1 | grep -v amdgpu.ids; fi
