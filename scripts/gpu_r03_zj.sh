#!/bin/bash
# C2 step: item pass workgroups per CU beyond 64 (one tile per workgroup at 128)
mkdir -p gpurun_out/r03_zj
timeout 400 python scripts/sweep_engine.py --steps 32 --warmup 8 --repeat 2 --out gpurun_out/r03_zj/item_grid.jsonl --configs \
  'item_grid_mult=128' 'item_grid_mult=96' 'item_grid_mult=48' 'item_grid_mult=256' 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-30s %s  %s' % (d['label'], ['%.4f' % x for x in d.get('ms_per_step_all', [])], {k: round(v, 4) for k, v in d.get('class_ms_per_step', {}).items()}))"
