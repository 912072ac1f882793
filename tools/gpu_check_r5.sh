#!/bin/bash
# Round 5 (same measurements as round 4's script, plus the MPC re-plan's host timeline): everything tools/gpu_check.sh measures (parity tests, smoke, bench, solver timelines, rocprof kernel traces,
# PMC passes - each PMC pass its own run with --kernel-trace only) plus what this round added: the accuracy table of all
# five configurations, fd_kernel by truncation / in-kernel stamps / SQ counters per phase / rocprof durations per phase,
# the table of all configurations, the solver beside a saturating neighbour.  Outputs under gpurun_out/ (summary/ for
# profiles/); tools/latency_model.py r04 (CPU) then condenses the model bench.py reports.
export ROUND=r05
bash tools/gpu_check.sh
R=r05
export TMPDIR=/tmp
( for m in "acrobot 40" "spinner 40" "hopper 50" "mini_cheetah 24 31 40" "allegro_hand 60"; do timeout 300 python tools/nd_accuracy.py $m; done ) 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_nd_accuracy.txt
timeout 300 python tools/fd_stops.py --both 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${R}_fd_phases.txt
if [ -f build/variants/stamps/libidto_hip.so ]; then
  IDTO_HIP_LIB=build/variants/stamps/libidto_hip.so timeout 120 python tools/fd_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${R}_fd_stamps.txt
fi
rm -f gpurun_out/fd_pmc.txt
timeout 600 bash tools/fd_pmc.sh mini_cheetah 40 "2 3 0" > /dev/null 2>&1
timeout 600 bash tools/fd_pmc.sh allegro_hand 60 "2 3 0" > /dev/null 2>&1
cp gpurun_out/fd_pmc.txt gpurun_out/${R}_fd_pmc.txt
{ timeout 300 bash tools/fd_rocprof_stops.sh mini_cheetah 40 "10 8 1 3 0" >/dev/null 2>&1; python - <<'PY'
import csv
for cfg, N in (("mini_cheetah", 40),):
    for s in (10, 8, 1, 3, 0):
        try:
            for r in csv.DictReader(open(f"gpurun_out/fdrp_{cfg}_{s}/r_kernel_stats.csv")):
                if "fd_kernel" in r["Name"]:
                    print(f"{cfg} N={N} fd_stop {s}: rocprofv3 average {float(r['AverageNs']) / 1e3:.2f} us over {r['Calls']} launches")
        except Exception as e:
            print(cfg, s, "missing", e)
PY
} | tee gpurun_out/${R}_fd_rocprof_phases.txt
timeout 600 python -m pytest tests/test_gpu_neighbour.py -m gpu -q -s 2>&1 | grep -v amdgpu.ids | tail -5 | tee gpurun_out/${R}_solver_beside_neighbour.txt
timeout 1500 bash tools/all_configs.sh > /dev/null 2>&1
timeout 60 ./tools/micro/launch_bench 2>&1 | tee gpurun_out/${R}_launch_bench.txt
# the equality-constrained iteration (hopper's YAML): kernels of the banded KKT step (csrc/kkt.h) and of the
# Schur-complement chain it replaced (option con_kkt = 0), per-kernel rocprofv3 averages
for kkt in 1 0; do
  IDTO_CON_KKT=$kkt bash tools/prof_full_iter.sh hopper 40 > /dev/null 2>&1
  cp gpurun_out/prof_fi_hopper/fi_kernel_stats.csv gpurun_out/${R}_constrained_iteration_hopper_kkt${kkt}_kernel_stats.csv
done
IDTO_CON_KKT=1 bash tools/prof_full_iter.sh allegro_hand 20 > /dev/null 2>&1
cp gpurun_out/prof_fi_allegro_hand/fi_kernel_stats.csv gpurun_out/${R}_constrained_iteration_allegro_kkt1_kernel_stats.csv
for kkt in 1 0; do for c in "acrobot 40" "spinner 40" "hopper 40" "allegro_hand 20"; do echo -n "IDTO_CON_KKT=$kkt  "; IDTO_CON_KKT=$kkt timeout 120 python tools/full_iter_prof.py $c 2>&1 | tail -1; done; done | tee -a gpurun_out/${R}_constrained_iteration_times.txt
# bit reproducibility of the constrained loop over fresh contexts (the KKT step's factorisations)
{ for c in "allegro_hand 60 60" "hopper 40 100" "spinner 40 60" "acrobot 40 60"; do timeout 600 python tools/stress_kkt.py $c 2>&1 | grep -v amdgpu.ids | tail -2; done; } | tee gpurun_out/${R}_kkt_stress.txt
timeout 600 python tools/fd_sweep.py 12 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/${R}_fd_sweep.txt
{ for c in "acrobot 40 1200" "spinner 40 1200" "acrobot 200 600"; do timeout 300 python tools/stress_solver.py $c 2>&1 | grep -v amdgpu.ids; done; } | tee gpurun_out/${R}_band_stress.txt
timeout 200 python tools/mpc_latency.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${R}_mpc_latency.txt
timeout 120 python tools/nd_timeline.py allegro_hand 60 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${R}_nd_timeline_allegro.txt
{ timeout 120 python tools/kkt_timeline.py allegro_hand 40; timeout 120 python tools/kkt_timeline.py allegro_hand 60; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${R}_kkt_timeline_allegro.txt
{ IDTO_TIMELINE_OPTS=nd_recursion=0 timeout 120 python tools/nd_timeline.py allegro_hand 60; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${R}_nd_timeline_allegro_rowwise_tail.txt
{ for c in "allegro_hand 60 20" "allegro_hand 40 20" "hopper 40 20"; do timeout 120 python tools/constrained_loop.py $c 2>&1 | grep -v amdgpu.ids | head -1; done; } | tee gpurun_out/${R}_constrained_loop_times.txt
timeout 300 bash tools/gpu_r5_batch32.sh 2>&1 | grep -v amdgpu.ids | head -4 | tee gpurun_out/${R}_allegro_batches_write_through.txt
timeout 300 python tools/mpc_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${R}_mpc_timeline.txt
timeout 300 python tools/band_phases.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${R}_band_phases.txt
ROUND=$R timeout 900 bash tools/prof_band.sh > /dev/null 2>&1
ls gpurun_out | head -80
