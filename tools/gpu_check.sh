#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, smoke, bench, rocprof kernel trace.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
timeout 600 python bench.py --steps 200 --warmup 20 2>&1 | tail -3 | tee gpurun_out/bench.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT && find gpurun_out/prof -type f | head -20; for f in $(find gpurun_out/prof -name "*kernel_stats*.csv" | head -1); do head -12 $f; done
