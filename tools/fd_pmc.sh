#!/bin/bash
# SQ counters of fd_kernel truncated after a phase (fd_stop): the difference between two stops is that phase's own
# instruction and wait counts.  usage: fd_pmc.sh <config> <N> "<stops>"  -> gpurun_out/fd_pmc.txt
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cfg=${1:-mini_cheetah}; N=${2:-40}; stops=${3:-"2 3 0"}
cd /tmp
for stop in $stops; do
  for pass in 1 2; do
    if [ $pass = 1 ]; then ctr="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU";
    else ctr="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INST_CYCLES_VMEM_RD"; fi
    d=$ROOT/gpurun_out/fdpmc_${cfg}_${stop}_${pass}
    rm -rf $d
    timeout -k 10 120 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -o r -- python $ROOT/tools/fd_pmc_driver.py $cfg $N $stop > /dev/null 2>&1
  done
done
cd $ROOT
python - "$cfg" "$stops" <<'PY' | tee -a gpurun_out/fd_pmc.txt
import csv, glob, collections, sys
cfg, stops = sys.argv[1], sys.argv[2].split()
for stop in stops:
    acc = collections.defaultdict(lambda: [0.0, 0])
    for p in (1, 2):
        for f in glob.glob(f"gpurun_out/fdpmc_{cfg}_{stop}_{p}/**/*counter_collection*.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "fd_kernel" not in row["Kernel_Name"]: continue
                a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    w = acc["SQ_WAVES"][0] / max(acc["SQ_WAVES"][1], 1)
    print(cfg, "fd_stop", stop, "waves/launch", round(w), {c: round(v[0] / v[1] / max(w, 1)) for c, v in sorted(acc.items()) if c != "SQ_WAVES"}, "(per wavefront)")
PY
