#!/bin/bash
# allegro batches on the seven-workgroup kernel: write-through publishing of the spike rows against plain stores + releasing fences
cd $GRAFT_REPO_ROOT
line() { timeout 600 python bench.py --config allegro_hand --num-steps 60 --batch 8 --no-full --no-cpu --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
b=json.loads(sys.stdin.read()); k=b['roofline']['all_kernels_avg_ms']
print(round(b['value']), {n: round(1e3*v,1) for n,v in k.items()}, [(e['problems'], round(e['value'])) for e in (b.get('batch_mode') or [])])
"; }
echo "default (write-through rows)"; line
echo "IDTO_ND_WT=2 (plain stores, releasing fence per wavefront)"; IDTO_ND_WT=2 line
