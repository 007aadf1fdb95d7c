#!/bin/bash
# user pass <UMODE 1> with the pairs' dL/dscore one per lane and the live rows two at a time: C3 and the adaptive-hinge launch path
mkdir -p gpurun_out/r03_zd
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_bench_parity.py -q -x -k "adaptive or c3 or matches_oracle or explicit" 2>&1 | tail -3 | tee gpurun_out/r03_zd/pytest.txt
for i in 1 2; do timeout 300 python bench.py --workload c3 --steps 32 --warmup 8 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'what': 'C3', 'ms_per_step': round(d['ms_per_step'], 4), 'M_interactions_per_s': round(d['value'] / 1e6, 1)}))" | tee -a gpurun_out/r03_zd/c3.jsonl; done
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03_zd/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 16 --warmup 4 > $GRAFT_REPO_ROOT/gpurun_out/r03_zd/prof_bench.json 2> /dev/null)
db=$(find gpurun_out/r03_zd/prof -name "*.db" | head -1)
[ -n "$db" ] && python scripts/summarize_prof.py "$db" gpurun_out/r03_zd/kernel_stats_c3.md "rocprofv3 --kernel-trace --stats -- python bench.py --workload c3 --steps 16 --warmup 4 (user pass <UMODE 1> + score pass batched)" gpurun_out/r03_zd/prof_bench.json && rm -rf gpurun_out/r03_zd/prof && head -10 gpurun_out/r03_zd/kernel_stats_c3.md | cut -c1-120
timeout 300 python scripts/bench_adaptive_small.py 2>/dev/null | grep '^{' | grep '"mid"' | tee gpurun_out/r03_zd/adaptive_launch_path.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['batch'], 'late' if d['late_item_sort'] else 'chunk-sorted', round(d['us_per_minibatch'], 1))"
