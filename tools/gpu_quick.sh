#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
timeout 120 python tools/solver_phases.py 2>&1 | tail -12 | tee gpurun_out/phases.log
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu 2>&1 | tail -2 | tee gpurun_out/bench.log
