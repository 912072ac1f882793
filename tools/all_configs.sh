#!/bin/bash
# bench line of every BASELINE configuration (one Gauss-Newton iteration = fd + assembly + factor/solve) -> gpurun_out/${R}_all_configs.txt
R=${ROUND:-r04}
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/${R}_all_configs.txt
echo "python bench.py --config <c> --num-steps <N> --batch <B> --no-full --steps 100 --warmup 10   (1x MI355X; HIP-event kernel times in us)" > $OUT
for cfg in "acrobot 40 16" "spinner 40 16" "hopper 50 16" "mini_cheetah 40 16" "allegro_hand 60 8"; do
  set -- $cfg
  timeout 600 python bench.py --config $1 --num-steps $2 --batch $3 --no-full --steps 100 --warmup 10 2>/dev/null | tail -1 > /tmp/line.json
  python - "$1" "$2" <<'PY' >> $OUT
import json, sys
b = json.loads(open('/tmp/line.json').read())
k = b['roofline']['all_kernels_avg_ms']
cpu = b.get('cpu_baseline', {}).get('iters_per_s_by_num_threads', {})
bm = b.get('batch_mode') or []
print(f"{sys.argv[1]} N={sys.argv[2]}: {b['value']:.0f} it/s, {1e3*b['ms_per_step']:.1f} us/step; kernels (us): "
      + ", ".join(f"{n} {1e3*v:.1f}" for n, v in k.items())
      + "; CPU port it/s by threads: " + ", ".join(f"{t}: {v:.0f}" for t, v in cpu.items())
      + "; batch: " + ", ".join(f"{e['problems']}: {e['value']:.0f}" for e in bm))
PY
done
cat $OUT
