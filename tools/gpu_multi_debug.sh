#!/bin/bash
# two ranks on the ONE GPU of the box (debugging aid: exercises bench.py's world > 1 code paths)
mkdir -p gpurun_out
export TMPDIR=/tmp
IDTO_BENCH_SAME_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 40 --warmup 5 2>&1 | tail -3 | cut -c1-3000 | tee gpurun_out/bench_2rank_same_gpu.log
timeout 300 python -m pytest tests/test_gpu_multi.py -q 2>&1 | tail -3
