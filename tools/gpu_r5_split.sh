#!/bin/bash
export TMPDIR=/tmp
for s in 0.46 0.52 0.58 0.64; do
  echo "split $s: $(IDTO_PIPE_SPLIT=$s timeout 200 python bench.py --steps 300 --warmup 30 --no-cpu --no-full --batch 0 2>&1 | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["all_kernels_avg_ms"])')"
done
