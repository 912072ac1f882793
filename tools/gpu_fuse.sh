#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
IDTO_TIMELINE_GN_STEP=1 timeout 200 python tools/nd_timeline.py 2>&1 | grep -v "^   pivots\|as follower\|back subst\|median" | tail -40 | tee gpurun_out/nd_timeline_fused.log
