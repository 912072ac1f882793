#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command (gpurun_out/prof_$R)
R=${ROUND:-r03}
cd /tmp; export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
B="python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu --no-full --batch 0"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$R -o $R -- $B > $ROOT/gpurun_out/prof_$R.log 2>&1
cd $ROOT
head -8 gpurun_out/prof_$R/${R}_kernel_stats.csv
