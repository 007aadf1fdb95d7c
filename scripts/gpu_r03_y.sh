#!/bin/bash
# item pass with every head's row early (k_item_pass<..., NPRE 4>, option item_lat_max_tiles): parity + same-box A/B
mkdir -p gpurun_out/r03_y
timeout 900 python -m pytest tests/test_gpu_engine.py -q -x -k "every_head_early" 2>&1 | tail -4 | tee gpurun_out/r03_y/pytest.txt
sum() { python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline']['kernels']
print(json.dumps({'what': '$1', 'item_lat_max_tiles': $2, 'us_per_minibatch': round(d['ms_per_step'] * 1e3, 2), 'kernel_avg_us': {a: round(b['avg_ms'] * 1e3, 2) for a, b in k.items()}}))"; }
for lat in 2048 0 2048 0; do
  for B in 4096 16384 65536; do
    timeout 300 python bench.py --batch $B --steps 256 --warmup 16 --no-cpu-baseline --no-probes --no-sharded-check --no-fit --no-overlapped --set item_lat_max_tiles=$lat 2>/dev/null | grep '^{' | sum "c2 tables, minibatch $B" $lat | tee -a gpurun_out/r03_y/item_lat_ab.jsonl
  done
  timeout 200 python bench.py --workload c4 --batch 256 --seq-len 10 --items 100000 --steps 400 --warmup 16 --set item_lat_max_tiles=$lat 2>/dev/null | grep '^{' | sum "poolnet 256 x 10" $lat | tee -a gpurun_out/r03_y/item_lat_ab.jsonl
  timeout 200 python bench.py --workload c4 --batch 256 --seq-len 200 --steps 200 --warmup 16 --set item_lat_max_tiles=$lat 2>/dev/null | grep '^{' | sum "poolnet 256 x 200" $lat | tee -a gpurun_out/r03_y/item_lat_ab.jsonl
done
