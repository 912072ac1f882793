#!/bin/bash
# round 4, first GPU call: parity of the straight-line evaluation + timings
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/fd_compare.py --oracle 2>&1 | tail -20 | tee gpurun_out/fd_compare.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kats.py tests/test_golden.py tests/test_gpu_fold.py tests/test_gpu_batch.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
