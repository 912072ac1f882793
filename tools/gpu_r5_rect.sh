#!/bin/bash
# the seven-workgroup kernel's back substitution in recursion form: tests, timelines, kernel durations
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_nd.py tests/test_gpu_solver_accuracy.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/rect_tests.txt
for o in 1 0; do
  echo "== nd_recursion=$o" >> gpurun_out/rect_timeline.txt
  IDTO_TIMELINE_OPTS="nd_recursion=$o" timeout 300 python tools/nd_timeline.py allegro_hand 60 >> gpurun_out/rect_timeline.txt 2>&1
done
echo "== cheetah (pipelined kernel)" >> gpurun_out/rect_timeline.txt
timeout 300 python tools/nd_timeline.py mini_cheetah 40 2>&1 | grep -E "recursion from|separator" >> gpurun_out/rect_timeline.txt
timeout 600 python -m pytest tests/test_gpu_pipe.py -x -q -m gpu 2>&1 | tail -3 >> gpurun_out/rect_tests.txt
