#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 800 python -m pytest tests/test_gpu_trust_region.py tests/test_gpu_optimizer.py -q 2>&1 | grep -a "passed\|failed\|FAILED\|Error\|assert" | head -40 | tee gpurun_out/pytest_tr.log
