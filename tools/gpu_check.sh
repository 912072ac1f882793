#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, smoke, bench, the solver timelines, rocprof kernel
# traces of the bench and of a full Solve, and the PMC passes (HBM traffic, SQ counters) - each PMC
# pass its own run with --kernel-trace only, never combined with other trace domains.
# Outputs under gpurun_out/; tools/pmc_summarize.py condenses them into gpurun_out/summary/ for profiles/.
set -x
R=${ROUND:-r04}
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/${R}_bench.json
timeout 120 python tools/nd_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${R}_nd_timeline.txt
IDTO_TIMELINE_GN_STEP=1 timeout 120 python tools/nd_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${R}_nd_timeline_gn_step.txt
{ timeout 60 ./tools/micro/xcd_pingpong; timeout 60 ./tools/micro/chain_bench; } 2>&1 | tee gpurun_out/${R}_microbench.txt
timeout 300 python tools/stress_solver.py mini_cheetah 40 600 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${R}_solver_stress.txt
timeout 120 python tools/solver_phases.py 2>&1 | grep -v amdgpu.ids | tail -28 | tee gpurun_out/${R}_solver_phases.txt
{ for c in "mini_cheetah 40" "hopper 40" "allegro_hand 20" "acrobot 40" "spinner 40"; do timeout 120 python tools/full_iter_prof.py $c 2>&1 | tail -1; done; } | tee gpurun_out/${R}_full_iteration_times.txt
ROOT=$GRAFT_REPO_ROOT
cd /tmp
B="python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu --no-full --batch 0"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof -o $R -- $B > $ROOT/gpurun_out/prof_bench.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_3launch -o $R -- $B --set asm_in_solver=0 > $ROOT/gpurun_out/prof_3launch.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_full -o ${R}_full -- python $ROOT/tools/full_iter_prof.py mini_cheetah 40 > $ROOT/gpurun_out/prof_full.log 2>&1
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_fetch -o $R -- $B > $ROOT/gpurun_out/pmc_fetch.log 2>&1
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_write -o $R -- $B > $ROOT/gpurun_out/pmc_write.log 2>&1
S="python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu --no-full --batch 0"
timeout -k 10 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_sq1 -o r -- $S > $ROOT/gpurun_out/pmc_sq1.log 2>&1
timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INST_CYCLES_VMEM_RD --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_sq3 -o r -- $S > $ROOT/gpurun_out/pmc_sq3.log 2>&1
cd $ROOT
python tools/pmc_summarize.py gpurun_out $R | tee gpurun_out/pmc_summary.log
