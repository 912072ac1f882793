#!/bin/bash
# rocprofv3 kernel durations of fd_kernel truncated at fd_stop (the kernel alone, without the HIP events' overhead)
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cfg=${1:-mini_cheetah}; N=${2:-40}; stops=${3:-"10 8 1 3 0"}
cd /tmp
for stop in $stops; do
  d=$ROOT/gpurun_out/fdrp_${cfg}_${stop}
  rm -rf $d
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- python $ROOT/tools/fd_pmc_driver.py $cfg $N $stop > /dev/null 2>&1
  f=$(find $d -name "*kernel_stats*.csv" | head -1)
  echo "$cfg fd_stop $stop: $(grep fd_kernel $f | head -1 | awk -F, '{print "calls", $2, "avg ns", $4}')"
done
