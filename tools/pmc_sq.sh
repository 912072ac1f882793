#!/bin/bash
# SQ counter passes (issue/wait breakdown, instruction cache) for the three kernels; separate
# --pmc runs with --kernel-trace only.  Output: gpurun_out/pmc_sq*/ + a per-kernel summary.
set -x
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_sq1 -o r -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu --no-full --batch 0 > $ROOT/gpurun_out/pmc_sq1.log 2>&1
timeout -k 10 240 rocprofv3 --pmc SQC_ICACHE_MISSES SQC_ICACHE_HITS SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_sq2 -o r -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu --no-full --batch 0 > $ROOT/gpurun_out/pmc_sq2.log 2>&1
timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INST_CYCLES_VMEM_RD --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_sq3 -o r -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu --no-full --batch 0 > $ROOT/gpurun_out/pmc_sq3.log 2>&1
cd $ROOT
python - <<'PY'
import csv, glob, collections
for d in ("pmc_sq1", "pmc_sq2", "pmc_sq3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection*.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1]
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    for k, cs in acc.items():
        if "rocclr" in k: continue
        print(d, k, {c: round(v[0] / v[1]) for c, v in cs.items()})
PY
