#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 800 python -m pytest tests/test_gpu_batch.py tests/test_gpu_trust_region.py tests/test_gpu_optimizer.py -q -x 2>&1 | grep -a "passed\|failed\|FAILED\|Error\|assert\|^E " | head -40 | tee gpurun_out/pytest_tr.log
