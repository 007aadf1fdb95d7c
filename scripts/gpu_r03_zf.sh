#!/bin/bash
# item pass: spilled runs fetched 4 instead of 2 records at a time (build variant sp4), C3 / C2 / C5 / C4, alternated
mkdir -p gpurun_out/r03_zf
for rep in 1 2; do for lib in default sp4; do
  if [ $lib = default ]; then unset SPOTLIGHT_HIP_LIB; else export SPOTLIGHT_HIP_LIB=$GRAFT_REPO_ROOT/spotlight_amd/csrc/ab/libspotlight_hip_$lib.so; fi
  for w in c3 c4; do timeout 300 python bench.py --workload $w --steps 32 --warmup 8 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'lib': '$lib', 'what': '$w', 'ms_per_step': round(d['ms_per_step'], 4)}))" | tee -a gpurun_out/r03_zf/ab.jsonl; done
  timeout 300 python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-probes --no-sharded-check --no-fit --no-overlapped 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline']['kernels']; print(json.dumps({'lib': '$lib', 'what': 'c2', 'ms_per_step': round(d['ms_per_step'], 4), 'item_pass_ms': round(k['item_pass']['avg_ms'], 4)}))" | tee -a gpurun_out/r03_zf/ab.jsonl
done; done
