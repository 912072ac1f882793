#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_batch.py -q -k "trust_region" 2>&1 | grep -a "passed\|failed\|FAILED\|Error\|^E " | head -30 | tee gpurun_out/pytest_one.log
