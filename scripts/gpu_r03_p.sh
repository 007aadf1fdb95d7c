#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-probes --no-sharded-check --no-fit > /dev/null 2>&1
for m in none none w9 none; do python scripts/probe_first_overlap.py $m 2>/dev/null | tail -1; done
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probes --no-sharded-check --no-fit 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench: %.4f ms  in-line(timers) %.4f  user %.4f item %.4f' % (d['ms_per_step'], r['in_line']['ms_per_step_with_kernel_timers'], r['kernels']['user_pass']['avg_ms'], r['kernels']['item_pass']['avg_ms']))"; done
timeout 600 python -m pytest tests/test_gpu_engine.py -q -k "pipelined or chunk or sampler" 2>&1 | tail -2
