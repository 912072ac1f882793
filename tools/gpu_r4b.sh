#!/bin/bash
# round 4: the whole GPU suite, the accuracy table, a bench line
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/pytest_gpu.log
( for m in "acrobot 40" "spinner 40" "hopper 50" "mini_cheetah 24 31 40" "allegro_hand 60"; do timeout 300 python tools/nd_accuracy.py $m; done ) > gpurun_out/nd_accuracy.txt 2>&1
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu --no-full --batch 0 2>&1 | tail -1 > gpurun_out/bench_quick.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_quick.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["all_kernels_avg_ms"])
PY
