#!/bin/bash
# ND / pipe solver timeline + accuracy against LU (development aid)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python tools/nd_timeline.py 2>&1 | tail -60 | tee gpurun_out/nd_timeline.log
timeout 200 python tools/nd_accuracy.py 2>&1 | tail -30 | tee gpurun_out/nd_accuracy.log
