#!/bin/bash
# PMC passes for the solver kernel of the bench (each pass its own run, --kernel-trace only)
cd /tmp; export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
S="python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu --no-full --batch 0"
P=${PMC:-"SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU"}
timeout -k 10 240 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_c -o r -- $S > $ROOT/gpurun_out/pmc_c.log 2>&1
cd $ROOT
python - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/pmc_c/**/*counter_collection.csv", recursive=True):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    for k in acc:
        if "pipe" in k or "fd_kernel" in k or "penta" in k:
            print(k, {c: round(v/cnt[(k,c)]) for c,v in acc[k].items()})
PY
tail -3 gpurun_out/pmc_c.log
