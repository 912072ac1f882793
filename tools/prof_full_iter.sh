#!/bin/bash
# rocprofv3 kernel stats of a full trust-region Solve (tools/full_iter_prof.py <model> <N>) -> gpurun_out/prof_fi_<model>
M=${1:-allegro_hand}; N=${2:-60}
cd /tmp; export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_fi_$M -o fi -- python $ROOT/tools/full_iter_prof.py $M $N > $ROOT/gpurun_out/prof_fi_$M.log 2>&1
cd $ROOT
tail -1 gpurun_out/prof_fi_$M.log
cut -d, -f1-4 gpurun_out/prof_fi_$M/fi_kernel_stats.csv | cut -c1-110 | head -24
