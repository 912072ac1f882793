#!/bin/bash
# solver tests + timeline + accuracy + short bench (development aid)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_nd.py tests/test_gpu_timeout.py tests/test_gpu_status.py tests/test_gpu_penta.py tests/test_gpu_fused.py -q 2>&1 | grep -a "passed\|failed\|FAILED" | tee gpurun_out/pytest_nd.log
timeout 200 python tools/nd_timeline.py 2>&1 | grep -v "^   pivots\|as follower" | tail -40 | tee gpurun_out/nd_timeline.log
timeout 200 python tools/nd_accuracy.py acrobot 40 63 2>&1 | tail -16 | tee gpurun_out/nd_accuracy_acrobot.log
timeout 200 python tools/nd_accuracy.py mini_cheetah 40 2>&1 | tail -8 | tee gpurun_out/nd_accuracy.log
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu 2>&1 | tail -1 | cut -c1-1400 | tee gpurun_out/bench.log
