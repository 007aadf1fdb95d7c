#!/bin/bash
# persistent kernels with the batched item phase (4 positions per row group per round, one-ahead record prefetch in the runs)
mkdir -p gpurun_out/r03_z
timeout 900 python -m pytest tests/test_gpu_engine.py -q -x -k "poolnet_epoch or epoch_kernel" 2>&1 | tail -3 | tee gpurun_out/r03_z/pytest_epoch.txt
for B in 256 1024 2048; do python bench.py --batch $B --steps 2000 --warmup 8 --no-cpu-baseline --no-probes --no-sharded-check --no-fit --set epoch_max_batch=2048 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'what': 'bpr, C2 tables, minibatch $B, persistent kernel', 'us_per_minibatch': round(d['ms_per_step']*1e3, 2)}))" | tee -a gpurun_out/r03_z/small_batches.jsonl; done
timeout 300 python scripts/bench_adaptive_small.py --routes 2>/dev/null | grep '^{' | grep persistent | tee -a gpurun_out/r03_z/small_batches.jsonl
for shape in "256 10" "256 16" "1024 4" "256 32"; do
  set -- $shape
  timeout 200 python bench.py --workload c4 --batch $1 --seq-len $2 --items 100000 --steps 400 --warmup 16 --set epoch_seq=1 --set epoch_seq_max_timesteps=1000000 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'what': 'poolnet $1 x $2, persistent kernel', 'us_per_minibatch': round(d['ms_per_step'] * 1e3, 2)}))" | tee -a gpurun_out/r03_z/small_batches.jsonl
done
