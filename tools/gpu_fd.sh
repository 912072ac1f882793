#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -a "passed\|failed\|FAILED\|Error" | head -8 | tee gpurun_out/pytest_gpu.log
timeout 120 python tools/fold_timing.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/fd_phases.txt
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu --no-full --batch 0 2>&1 | tail -1 | cut -c1-200
