#!/bin/bash
# bench line of several library variants (build/variants/<name>/libidto_hip.so), usage: gpu_r5_variants.sh name ...
export TMPDIR=/tmp
for v in "$@"; do
  L=build/variants/$v/libidto_hip.so; [ "$v" = base ] && L=idto_amd/libidto_hip.so
  for rep in 1 2; do
  echo "$v: $(IDTO_HIP_LIB=$L timeout 200 python bench.py --steps 300 --warmup 30 --no-cpu --no-full --batch 0 2>&1 | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"]), d["roofline"]["all_kernels_avg_ms"])')"
  done
done
