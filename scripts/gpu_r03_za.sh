#!/bin/bash
# anatomy of the persistent kernels: time per minibatch with phases skipped (option epoch_debug; results are garbage, times are not)
mkdir -p gpurun_out/r03_za
for dbg in 0 1 8 16 24 2; do
  timeout 200 python bench.py --workload c4 --batch 256 --seq-len 10 --items 100000 --steps 400 --warmup 16 --set epoch_seq=1 --set epoch_debug=$dbg --no-loss-check 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'what': 'poolnet 256 x 10, persistent kernel', 'epoch_debug': $dbg, 'us_per_minibatch': round(d['ms_per_step'] * 1e3, 2)}))" | tee -a gpurun_out/r03_za/anatomy.jsonl
done
for dbg in 0 1 8 16 32 56 2; do
  timeout 200 python scripts/bench_adaptive_small.py --routes --set=epoch_debug=$dbg 2>/dev/null | grep '^{' | grep persistent | grep '"batch": \(256\|1024\),' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(json.dumps({'what': 'adaptive hinge (5 draws), %s shape, minibatch %d, persistent kernel' % (d['shape'], d['batch']), 'epoch_debug': $dbg, 'us_per_minibatch': round(d['us_per_minibatch'], 2)}))" | tee -a gpurun_out/r03_za/anatomy.jsonl
done
