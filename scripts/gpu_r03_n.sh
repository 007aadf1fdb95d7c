#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03o; mkdir -p $O
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-probes --no-sharded-check --no-fit > /dev/null 2>&1
python scripts/sweep_engine.py --steps 20 --warmup 5 --repeat 3 --out $O/c2_K20.jsonl --configs overlap_prep=0 > $O/c2_K20.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r03o/c2_K20.jsonl'):
    d=json.loads(l); print(d['label'], [round(x,4) for x in d['ms_per_step_all']])
PY
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probes --no-sharded-check --no-fit 2>/dev/null | tee -a $O/bench3.jsonl | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench: %.4f ms  in-line(timers) %.4f  user %.4f item %.4f' % (d['ms_per_step'], r['in_line']['ms_per_step_with_kernel_timers'], r['kernels']['user_pass']['avg_ms'], r['kernels']['item_pass']['avg_ms']))"; done
