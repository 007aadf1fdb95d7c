#!/bin/bash
# kernel stats of the bench region alone (no fit_end_to_end leg)
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02_final2; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-probes --no-sharded-check --no-fit > $OUT/prof_bench.json 2> $OUT/prof.err)
db=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$db" ] && python scripts/summarize_prof.py "$db" $OUT/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-probes --no-sharded-check --no-fit (r02_final2)" $OUT/prof_bench.json && rm -rf $OUT/prof && head -14 $OUT/kernel_stats.md | cut -c1-120
