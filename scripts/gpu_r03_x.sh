#!/bin/bash
# PoolNet on the persistent route: bit-identity on the GPU + time per minibatch against the launches
mkdir -p gpurun_out/r03_x
timeout 900 python -m pytest tests/test_gpu_engine.py -q -x -k "poolnet_epoch or epoch_kernel" 2>&1 | tail -5 | tee gpurun_out/r03_x/pytest_epoch.txt
for shape in "256 10" "256 16" "1024 4" "256 32" "256 200"; do
  set -- $shape
  for r in 1 0; do
    timeout 200 python bench.py --workload c4 --batch $1 --seq-len $2 --items 100000 --steps 400 --warmup 16 --set epoch_seq=$r --set epoch_seq_max_timesteps=1000000 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline']['kernels']
print(json.dumps({'sequences': $1, 'seq_len': $2, 'route': 'persistent kernel' if $r else 'launches', 'us_per_minibatch': round(d['ms_per_step'] * 1e3, 2), 'timesteps_per_s': d['value'], 'kernel_avg_ms': {a: round(b['avg_ms'], 5) for a, b in k.items()}, 'other': d['roofline']['other_ms_per_step']}))" | tee -a gpurun_out/r03_x/poolnet_routes.jsonl
  done
done
