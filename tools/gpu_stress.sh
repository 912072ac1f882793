#!/bin/bash
mkdir -p gpurun_out
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_gpu_nd.py -x -q 2>&1 | grep -a "passed\|failed\|FAILED\|Error" | head -5; done | tee gpurun_out/stress.log
