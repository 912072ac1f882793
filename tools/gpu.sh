#!/bin/bash
# Everything that runs on the GPU box (through gpurun), one script, one sub-command per job:
#
#   gpurun --timeout 1500 -- 'bash tools/gpu.sh tests [pytest args]'   the -m gpu suite (or the named files) + smoke
#   ... 'bash tools/gpu.sh quick [tag]'          solver timelines (alone / inside the step) + a short bench line
#   ... 'bash tools/gpu.sh solver [tag]'         the solvers' tests, then `quick`
#   ... 'bash tools/gpu.sh variants name ...'    bench line (twice) + timeline of build/variants/<name>/libidto_hip.so (`base`: the tree's)
#   ... 'bash tools/gpu.sh stress'               bit reproducibility over fresh contexts: every solver family, both release protocols
#   ... 'bash tools/gpu.sh batch32'              allegro batches, write-through rows against releasing fences
#   ... 'bash tools/gpu.sh mpc [tag]'            one MPC re-plan mark by mark + its latency
#   ... 'bash tools/gpu.sh prof [cmd...]'        rocprofv3 --kernel-trace --stats of the bench command (or of cmd)
#   ... 'bash tools/gpu.sh two-ranks'            bench.py's world > 1 paths with both ranks on the box's one GPU
#   ... 'ROUND=r06 bash tools/gpu.sh check'      the round's full measurement set -> gpurun_out/ (summary/ for profiles/)
#
# PMC passes are runs of their own with --kernel-trace only (never combined with other trace domains).
# Rounds 1-5 kept one script per lease (tools/gpu_r4a.sh ... gpu_r5_variants.sh, gpu_check*.sh); this file replaces them.
R=${ROUND:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
quiet() { grep -v amdgpu.ids; }
BENCH_SHORT="python bench.py --steps 200 --warmup 20 --no-cpu"
BENCH_KERNELS="python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu --no-full --batch 0"
bench_line() { python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"]), d["ms_per_step"], d["roofline"]["all_kernels_avg_ms"])'; }

cmd=${1:-check}; shift || true
case "$cmd" in
tests)
  timeout 1700 python -m pytest ${@:-tests} -m gpu -x -q 2>&1 | quiet | tail -15 | tee gpurun_out/${R}_pytest_gpu.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | quiet | tail -3 | tee gpurun_out/${R}_smoke.txt
  ;;
quick)
  T=${1:-x}
  timeout 120 python tools/nd_timeline.py 2>&1 | quiet | tee gpurun_out/q_${T}_nd_timeline.txt | grep -v "pivots 4\|row 4 as\|median"
  IDTO_TIMELINE_GN_STEP=1 timeout 120 python tools/nd_timeline.py 2>&1 | quiet > gpurun_out/q_${T}_nd_timeline_gn.txt
  timeout 300 $BENCH_SHORT 2>&1 | tail -1 > gpurun_out/q_${T}_bench.json
  echo -n "bench: "; bench_line < gpurun_out/q_${T}_bench.json
  ;;
solver)
  timeout 900 python -m pytest tests/test_gpu_nd.py tests/test_gpu_solver_accuracy.py tests/test_gpu_penta.py tests/test_gpu_timeout.py \
    tests/test_gpu_fold.py tests/test_gpu_batch.py tests/test_gpu_band.py -m gpu -x -q 2>&1 | quiet | tail -8 | tee gpurun_out/s_${1:-x}_pytest.txt
  bash tools/gpu.sh quick "${1:-x}"
  ;;
variants)
  for v in "$@"; do
    L=build/variants/$v/libidto_hip.so; [ "$v" = base ] && L=idto_amd/libidto_hip.so
    for rep in 1 2; do echo "$v: $(IDTO_HIP_LIB=$L timeout 200 python bench.py --steps 300 --warmup 30 --no-cpu --no-full --batch 0 2>&1 | tail -1 | bench_line)"; done
    IDTO_HIP_LIB=$L timeout 120 python tools/nd_timeline.py 2>&1 | grep "elimination of row\|^separator \|joiner \|producer "
  done
  ;;
stress)
  # (the row hand-overs are ordered by write-through stores + s_waitcnt, penta_nd.h release_row; IDTO_ND_WT=2 is the
  # formal release.  Both protocols, every multi-workgroup family - pipelined K = 19, seven workgroups K = 23 and the KKT
  # systems' K = 29, the band kernel - over fresh contexts, bits compared.)
  { for c in "mini_cheetah 40 600" "allegro_hand 60 300" "acrobot 40 600" "spinner 40 600"; do
      timeout 600 python tools/stress_solver.py $c 2>&1 | quiet | tail -3
      IDTO_ND_WT=2 timeout 600 python tools/stress_solver.py $c 2>&1 | quiet | tail -1 | sed 's/^/IDTO_ND_WT=2  /'
    done
    for c in "allegro_hand 60 400" "mini_cheetah 40 400" "mini_cheetah 24 400"; do timeout 600 python tools/nd_stress.py $c 2>&1 | quiet | tail -2; done
    for c in "allegro_hand 60 60" "allegro_hand 40 60" "hopper 40 100"; do timeout 600 python tools/stress_kkt.py $c 2>&1 | quiet | tail -2; done
    # (the resident trust-region loop: tr_iter_kernel's hand-over between its workgroups, the small models' one-launch iteration)
    for c in "mini_cheetah 40 8 300" "allegro_hand 60 3 150" "hopper 40 10 300 c" "acrobot 40 20 300 c" "spinner 40 12 300" "allegro_hand 20 3 150 c"; do timeout 600 python tools/stress_tr.py $c 2>&1 | quiet | tail -1; done
  } | tee gpurun_out/${R}_solver_stress.txt
  ;;
batch32)
  line() { timeout 600 python bench.py --config allegro_hand --num-steps 60 --batch 8 --no-full --no-cpu --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
b=json.loads(sys.stdin.read()); k=b['roofline']['all_kernels_avg_ms']
print(round(b['value']), {n: round(1e3*v,1) for n,v in k.items()}, [(e['problems'], round(e['value'])) for e in (b.get('batch_mode') or [])])"; }
  { echo "default (write-through rows)"; line; echo "IDTO_ND_WT=2 (plain stores, releasing fence per wavefront)"; IDTO_ND_WT=2 line; } | tee gpurun_out/${R}_allegro_batches_write_through.txt
  ;;
mpc)
  timeout 300 python tools/mpc_timeline.py 2>&1 | quiet | tee gpurun_out/${R}_mpc_timeline.txt
  timeout 200 python tools/mpc_latency.py 2>&1 | quiet | tee gpurun_out/${R}_mpc_latency.txt
  ;;
prof)
  C=${@:-$BENCH_KERNELS}
  cd /tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$R -o $R -- $C > $ROOT/gpurun_out/prof_$R.log 2>&1
  cd $ROOT
  head -8 gpurun_out/prof_$R/${R}_kernel_stats.csv; tail -1 gpurun_out/prof_$R.log | cut -c1-400
  ;;
two-ranks)
  IDTO_BENCH_SAME_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 --steps 40 --warmup 5 2>&1 | tail -3 | cut -c1-3000 | tee gpurun_out/bench_2rank_same_gpu.log
  timeout 300 python -m pytest tests/test_gpu_multi.py -q 2>&1 | tail -3
  ;;
check)
  set -x
  rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4
  timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | quiet | tail -25 | tee gpurun_out/${R}_pytest_gpu.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | quiet | tail -5 | tee gpurun_out/${R}_smoke.txt
  timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/${R}_bench.json
  timeout 120 python tools/nd_timeline.py 2>&1 | quiet | tee gpurun_out/${R}_nd_timeline.txt
  IDTO_TIMELINE_GN_STEP=1 timeout 120 python tools/nd_timeline.py 2>&1 | quiet | tee gpurun_out/${R}_nd_timeline_gn_step.txt
  timeout 120 python tools/nd_timeline.py allegro_hand 60 2>&1 | quiet | tee gpurun_out/${R}_nd_timeline_allegro.txt
  { timeout 120 python tools/kkt_timeline.py allegro_hand 40; timeout 120 python tools/kkt_timeline.py allegro_hand 60; } 2>&1 | quiet | tee gpurun_out/${R}_kkt_timeline_allegro.txt
  { timeout 60 ./tools/micro/xcd_pingpong; timeout 60 ./tools/micro/chain_bench; timeout 60 ./tools/micro/launch_bench; } 2>&1 | tee gpurun_out/${R}_microbench.txt
  timeout 120 python tools/solver_phases.py 2>&1 | quiet | tail -28 | tee gpurun_out/${R}_solver_phases.txt
  { for c in "mini_cheetah 40" "hopper 40" "allegro_hand 20" "acrobot 40" "spinner 40"; do timeout 120 python tools/full_iter_prof.py $c 2>&1 | tail -1; done; } | tee gpurun_out/${R}_full_iteration_times.txt
  # kernel traces: the bench as shipped (two launches per step), the assembly in a launch of its own, a full Solve
  cd /tmp
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof -o $R -- $BENCH_KERNELS > $ROOT/gpurun_out/prof_bench.log 2>&1
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_3launch -o $R -- $BENCH_KERNELS --set asm_in_solver=0 > $ROOT/gpurun_out/prof_3launch.log 2>&1
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_full -o ${R}_full -- python $ROOT/tools/full_iter_prof.py mini_cheetah 40 > $ROOT/gpurun_out/prof_full.log 2>&1
  # counters, one pass each
  timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_fetch -o $R -- $BENCH_KERNELS > $ROOT/gpurun_out/pmc_fetch.log 2>&1
  timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_write -o $R -- $BENCH_KERNELS > $ROOT/gpurun_out/pmc_write.log 2>&1
  S="python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu --no-full --batch 0"
  timeout -k 10 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_sq1 -o r -- $S > $ROOT/gpurun_out/pmc_sq1.log 2>&1
  timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INST_CYCLES_VMEM_RD --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_sq3 -o r -- $S > $ROOT/gpurun_out/pmc_sq3.log 2>&1
  cd $ROOT
  python tools/pmc_summarize.py gpurun_out $R | tee gpurun_out/pmc_summary.log
  # accuracy table of all five configurations; fd_kernel by truncation, in-kernel stamps, SQ counters and rocprof durations per phase
  ( for m in "acrobot 40" "spinner 40" "hopper 50" "mini_cheetah 24 31 40" "allegro_hand 60"; do timeout 300 python tools/nd_accuracy.py $m; done ) 2>&1 | quiet > gpurun_out/${R}_nd_accuracy.txt
  timeout 300 python tools/fd_stops.py --both 2>&1 | quiet | tee gpurun_out/${R}_fd_phases.txt
  if [ -f build/variants/stamps/libidto_hip.so ]; then
    IDTO_HIP_LIB=build/variants/stamps/libidto_hip.so timeout 120 python tools/fd_stamps.py 2>&1 | quiet | tee gpurun_out/${R}_fd_stamps.txt
  fi
  rm -f gpurun_out/fd_pmc.txt
  timeout 600 bash tools/fd_pmc.sh mini_cheetah 40 "2 3 0" > /dev/null 2>&1
  timeout 600 bash tools/fd_pmc.sh allegro_hand 60 "2 3 0" > /dev/null 2>&1
  cp gpurun_out/fd_pmc.txt gpurun_out/${R}_fd_pmc.txt
  timeout 1500 bash tools/all_configs.sh > /dev/null 2>&1
  # the equality-constrained iteration: kernels of the banded KKT step (csrc/kkt.h) and of the Schur-complement chain (con_kkt = 0)
  for kkt in 1 0; do
    IDTO_CON_KKT=$kkt bash tools/prof_full_iter.sh hopper 40 > /dev/null 2>&1
    cp gpurun_out/prof_fi_hopper/fi_kernel_stats.csv gpurun_out/${R}_constrained_iteration_hopper_kkt${kkt}_kernel_stats.csv
  done
  for kkt in 1 0; do for c in "acrobot 40" "spinner 40" "hopper 40" "allegro_hand 20"; do echo -n "IDTO_CON_KKT=$kkt  "; IDTO_CON_KKT=$kkt timeout 120 python tools/full_iter_prof.py $c 2>&1 | tail -1; done; done | tee gpurun_out/${R}_constrained_iteration_times.txt
  { for c in "allegro_hand 60 20" "allegro_hand 40 20" "hopper 40 20"; do timeout 120 python tools/constrained_loop.py $c 2>&1 | quiet | head -1; done; } | tee gpurun_out/${R}_constrained_loop_times.txt
  timeout 600 python tools/fd_sweep.py 12 2>&1 | quiet | tail -3 | tee gpurun_out/${R}_fd_sweep.txt
  bash tools/gpu.sh stress
  bash tools/gpu.sh mpc
  bash tools/gpu.sh batch32
  if [ -f build/variants/trstamps/libidto_hip.so ]; then   # (bash tools/main_variants.sh trstamps -DIDTO_TR_STAMPS)
    for m in mini_cheetah acrobot allegro_hand; do IDTO_HIP_LIB=build/variants/trstamps/libidto_hip.so timeout 200 python tools/tr_stamps.py $m 2>&1 | quiet; done | tee gpurun_out/${R}_tr_stamps.txt
  fi
  timeout 300 python tools/band_phases.py 2>&1 | quiet | tee gpurun_out/${R}_band_phases.txt
  timeout 300 python tools/small_iter_time.py 2>&1 | quiet | tee gpurun_out/${R}_small_iteration_times.txt
  IDTO_SMALL_STAMPS=1 timeout 120 python tools/small_phases.py 2>&1 | quiet | grep "phases\|inside" | tee gpurun_out/${R}_small_phases.txt
  ROUND=$R timeout 900 bash tools/prof_band.sh > /dev/null 2>&1
  timeout 600 python -m pytest tests/test_gpu_neighbour.py -m "gpu or timing" -q -s 2>&1 | quiet | tail -8 | tee gpurun_out/${R}_solver_beside_neighbour.txt
  ls gpurun_out | head -100
  ;;
*) echo "unknown sub-command $cmd (see the header of tools/gpu.sh)"; exit 2;;
esac
