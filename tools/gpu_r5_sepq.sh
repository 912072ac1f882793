#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_nd.py tests/test_gpu_solver_accuracy.py tests/test_gpu_batch.py tests/test_gpu_trust_region.py tests/test_gpu_penta.py -x -q -m gpu 2>&1 | tail -2
python tools/kkt_timeline.py allegro_hand 40 2>&1 | grep "^separator\|last row of\|producer\|joiner " | cut -c1-200
IDTO_TIMELINE_OPTS="nd_recursion=1" python tools/nd_timeline.py allegro_hand 60 2>&1 | grep "producer \|last row\|^separator  "
python tools/nd_timeline.py mini_cheetah 40 2>&1 | grep "producer \|joiner \|^separator  "
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu 2>&1 | tail -1 | cut -c1-400
