#!/bin/bash
# rocprofv3 of the small models' step (fd_kernel + penta_band_kernel, csrc/penta_band.h): kernel trace + stats for
# acrobot and spinner, and one SQ counter pass (issue / wait breakdown; its own run, --kernel-trace only) for acrobot.
# Output: gpurun_out/${ROUND}_band_kernel_stats.txt
R=${ROUND:-r04}
ROOT=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for cfg in acrobot spinner; do
  B="python $ROOT/bench.py --config $cfg --num-steps 40 --steps 200 --warmup 20 --no-cpu --no-full --batch 0"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_band_$cfg -o b -- $B > $ROOT/gpurun_out/prof_band_$cfg.log 2>&1
done
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_band -o r -- python $ROOT/bench.py --config acrobot --num-steps 40 --steps 20 --warmup 3 --no-cpu --no-full --batch 0 > $ROOT/gpurun_out/pmc_band.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_band2 -o r -- python $ROOT/bench.py --config acrobot --num-steps 40 --steps 20 --warmup 3 --no-cpu --no-full --batch 0 > $ROOT/gpurun_out/pmc_band2.log 2>&1
cd $ROOT
python - <<'PY' | tee gpurun_out/${R}_band_kernel_stats.txt
import csv, glob, collections
for cfg in ("acrobot", "spinner"):
    print(f"{cfg} N=40: rocprofv3 --kernel-trace --stats -- python bench.py --config {cfg} --num-steps 40 --steps 200 --warmup 20 --no-cpu --no-full --batch 0")
    for f in glob.glob(f"gpurun_out/prof_band_{cfg}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "rocclr" in r["Name"]: continue
            print("   %-70s calls %5s  average %7.2f us  (min %.2f, max %.2f)" % (r["Name"].replace("idto_dev::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
print("acrobot N=40, SQ counters per launch (separate --pmc runs, --kernel-trace only; summed over the launch's wavefronts):")
for d in ("pmc_band", "pmc_band2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection*.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("idto_dev::", "")
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    for k, cs in acc.items():
        if "rocclr" in k: continue
        print("   ", k, {c: round(v[0] / v[1]) for c, v in cs.items()})
PY
