#!/bin/bash
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_mpc
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_mpc -o m -- python $GRAFT_REPO_ROOT/tools/mpc_latency.py ${1:-mini_cheetah} > $GRAFT_REPO_ROOT/gpurun_out/prof_mpc.log 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_mpc/**/m_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
