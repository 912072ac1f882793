#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, smoke, bench, rocprof kernel trace and the
# two PMC passes for HBM traffic (separate runs; never combined with other trace domains).
set -x
R=${ROUND:-r01}
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
timeout 600 python bench.py --steps 200 --warmup 20 2>&1 | tail -3 | tee gpurun_out/bench.log
timeout 120 python tools/solver_phases.py 2>&1 | tail -14 | tee gpurun_out/phases.log
ROOT=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof -o $R -- python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu --no-full --batch 0 > $ROOT/gpurun_out/prof_bench.log 2>&1
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_fetch -o $R -- python $ROOT/bench.py --steps 50 --warmup 5 --no-cpu --no-full --batch 0 > $ROOT/gpurun_out/pmc_fetch.log 2>&1
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_write -o $R -- python $ROOT/bench.py --steps 50 --warmup 5 --no-cpu --no-full --batch 0 > $ROOT/gpurun_out/pmc_write.log 2>&1
cd $ROOT
find gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write -type f | head -30
for f in $(find gpurun_out/prof -name "*kernel_stats*.csv" | head -1); do head -12 $f; done
for f in $(find gpurun_out/pmc_fetch -name "*counter_collection*.csv" | head -1); do head -5 $f; done
python tools/pmc_summarize.py gpurun_out $R | tee gpurun_out/pmc_summary.log
