#!/bin/bash
# minibatch 65 536 on the C2 tables: grid multipliers of the two passes
mkdir -p gpurun_out/r03_zi
timeout 500 python scripts/sweep_engine.py --batch 65536 --steps 256 --warmup 16 --repeat 2 --out gpurun_out/r03_zi/grid_65536.jsonl --configs \
  'user_grid_mult=16' 'user_grid_mult=4' 'user_grid_mult=12' 'user_lat_max_batch=65536' 'user_lat_max_batch=65536,user_grid_mult=16' 'item_lat_max_tiles=0' 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-50s %s  %s' % (d['label'], ['%.4f' % x for x in d.get('ms_per_step_all', [])], {k: round(v * 1e3, 1) for k, v in d.get('class_ms_per_step', {}).items()}))"
