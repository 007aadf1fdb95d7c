#!/bin/bash
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02_v; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python scripts/bench_fit.py 100000000 2>$OUT/fit.err | tee $OUT/bench_fit_1e8.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('shuffle %.1f ms, fit %.1f ms/epoch = %.3f G/s' % (d['device_shuffle_s']*1e3, d['fit_s_per_epoch']*1e3, d['fit_interactions_per_s']/1e9))"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o fit -- python $R/scripts/bench_fit.py 30000000 > $OUT/prof_fit.json 2> $OUT/prof.err)
db=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$db" ] && python scripts/summarize_prof.py "$db" $OUT/fit_kernel_stats_3e7.md "rocprofv3 --kernel-trace --stats -- python scripts/bench_fit.py 30000000" && rm -rf $OUT/prof && head -30 $OUT/fit_kernel_stats_3e7.md | cut -c1-110
