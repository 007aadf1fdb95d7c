#!/bin/bash
# persistent kernel, pair losses: a user's further occurrences one step ahead (rows requested before the current one is scored)
mkdir -p gpurun_out/r03_zg
timeout 600 python -m pytest tests/test_gpu_engine.py -q -x -k "epoch_kernel" 2>&1 | tail -2 | tee gpurun_out/r03_zg/pytest.txt
for shape in "943 1682 32" "10000000 1000000 64"; do set -- $shape
for B in 256 1024; do for loss in bpr; do
python bench.py --users $1 --items $2 --dim $3 --batch $B --loss $loss --steps 2000 --warmup 8 --no-cpu-baseline --no-probes --no-sharded-check --no-fit 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'what': '$loss, $1 x $2 dim $3, minibatch $B, persistent kernel', 'us_per_minibatch': round(d['ms_per_step']*1e3, 2)}))" | tee -a gpurun_out/r03_zg/small_batches.jsonl
done; done; done
