#!/bin/bash
mkdir -p gpurun_out
timeout 120 ./tools/micro/xcd_pingpong 2>&1 | tee gpurun_out/xcd_pingpong.log
timeout 200 python tools/nd_timeline.py 2>&1 | grep -v "^   pivots\|as follower\|elimination of" | tail -40 | tee gpurun_out/nd_timeline.log
