#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_fold.py tests/test_gpu_nd.py tests/test_gpu_timeout.py tests/test_gpu_parity.py -x -q 2>&1 | grep -a "passed\|failed\|FAILED\|Error" | head -6 | tee gpurun_out/pytest_fuse.log
IDTO_TIMELINE_GN_STEP=1 timeout 200 python tools/nd_timeline.py 2>&1 | grep -v "^   pivots\|as follower\|back subst\|median\|amdgpu\|first join" | cut -c1-250 | tail -12 | tee gpurun_out/nd_timeline_fused.log
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu --no-full --batch 0 2>&1 | tail -1 | cut -c1-200 | tee gpurun_out/bench.log
timeout 300 python tools/stress_solver.py mini_cheetah 40 300 2>&1 | grep -v amdgpu
