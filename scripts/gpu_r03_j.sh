#!/bin/bash
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03k; mkdir -p $O; export TMPDIR=/tmp
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-probes --no-sharded-check --no-fit > /dev/null 2>&1   # the box's first process runs slow
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --user-zipf 1.0 --item-zipf 1.0 --steps 8 --warmup 2 --no-cpu-baseline --no-probes --no-sharded-check --no-fit > $O/prof_zipf.json 2> $O/prof.err)
db=$(find $O/prof -name "*.db" | head -1)
[ -n "$db" ] && python scripts/summarize_prof.py "$db" $O/kernel_stats_zipf.md "rocprofv3 --kernel-trace --stats -- python bench.py --user-zipf 1.0 --item-zipf 1.0 --steps 8 --warmup 2 (r03k)" $O/prof_zipf.json && rm -rf $O/prof && head -12 $O/kernel_stats_zipf.md
for z in 1.0 1.2; do python bench.py --user-zipf $z --item-zipf $z --steps 8 --warmup 2 --no-cpu-baseline --no-probes --no-sharded-check --no-fit 2>/dev/null | tee $O/bench_bothzipf_$z.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('both Zipf($z): %.3f G/s, user pass %.3f ms, item pass %.3f ms' % (d['value']/1e9, r['kernels']['user_pass']['avg_ms'], r['kernels']['item_pass']['avg_ms']))"; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probes --no-sharded-check --no-fit 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('uniform: %.3f G/s %.4f ms' % (d['value']/1e9, d['ms_per_step']))"
