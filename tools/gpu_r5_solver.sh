#!/bin/bash
# Round 5 solver loop: the solver's tests, then its timeline and a short bench line.  usage: gpu_r5_solver.sh [tag]
T=${1:-x}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_nd.py tests/test_gpu_solver_accuracy.py tests/test_gpu_penta.py tests/test_gpu_timeout.py tests/test_gpu_fold.py tests/test_gpu_batch.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -8 | tee gpurun_out/r5s_${T}_pytest.txt
timeout 120 python tools/nd_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5s_${T}_timeline.txt | grep -v "pivots 4\|row 4 as\|median" 
IDTO_TIMELINE_GN_STEP=1 timeout 120 python tools/nd_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5s_${T}_timeline_gn.txt
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu 2>&1 | tail -1 > gpurun_out/r5s_${T}_bench.json
python - <<PY
import json
d = json.load(open("gpurun_out/r5s_${T}_bench.json"))
print("bench", d["value"], d["ms_per_step"], d["roofline"]["all_kernels_avg_ms"])
PY
