#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for sp in 0.575 0.52 0.47; do
echo "== share $sp"
IDTO_PIPE_SPLIT=$sp timeout 200 python tools/nd_timeline.py 2>&1 | grep -v "^   pivots\|as follower\|back subst\|median" | tail -9
IDTO_PIPE_SPLIT=$sp timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu --no-full --batch 0 2>&1 | tail -1 | cut -c1-200
done 2>&1 | tee gpurun_out/split.log
