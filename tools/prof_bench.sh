#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command (gpurun_out/prof_$R), as shipped (two launches per step: the
# solver's grid assembles g and H) and with the assembly in its own launch (asm_in_solver=0: the solver alone)
R=${ROUND:-r03}
cd /tmp; export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
B="python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu --no-full --batch 0"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$R -o $R -- $B > $ROOT/gpurun_out/prof_$R.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_${R}_3launch -o $R -- $B --set asm_in_solver=0 > $ROOT/gpurun_out/prof_${R}_3launch.log 2>&1
cd $ROOT
head -8 gpurun_out/prof_$R/${R}_kernel_stats.csv
head -8 gpurun_out/prof_${R}_3launch/${R}_kernel_stats.csv
tail -1 gpurun_out/prof_$R.log | cut -c1-400
tail -1 gpurun_out/prof_${R}_3launch.log | cut -c1-400
