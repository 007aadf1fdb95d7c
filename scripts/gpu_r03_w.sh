#!/bin/bash
# overlapped prep with a short first chunk (option first_chunk), calls of 20 and 32 minibatches at the C2 shape
mkdir -p gpurun_out/r03_w
for K in 20 32; do
timeout 400 python scripts/sweep_engine.py --steps $K --warmup 5 --repeat 3 --out gpurun_out/r03_w/first_chunk_K$K.jsonl --configs \
  'overlap_prep=1' 'overlap_prep=1,first_chunk=1' 'overlap_prep=1,first_chunk=2' 'overlap_prep=1,first_chunk=3' 'overlap_prep=1,first_chunk=4' \
  'overlap_prep=1,first_chunk=2,chunk_interactions=4194304' 'overlap_prep=1,first_chunk=2,user_grid_mult=8' 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-60s %s  %s' % (d['label'], ['%.4f' % x for x in d.get('ms_per_step_all', [])], {k: round(v, 4) for k, v in d.get('class_ms_per_step', {}).items()}))"
done
