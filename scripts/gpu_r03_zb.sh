#!/bin/bash
# counter-based account of C3 (adaptive hinge + BloomEmbedding item table) and C4 (PoolNet): kernel trace + PMC traffic per kernel
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
for w in c3 c4; do
  OUT=$R/gpurun_out/r03_zb_$w; mkdir -p $OUT
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $R/bench.py --workload $w --steps 16 --warmup 4 > $OUT/prof_bench.json 2> $OUT/prof.err)
  db=$(find $OUT/prof -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/summarize_prof.py "$db" $OUT/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --workload $w --steps 16 --warmup 4" $OUT/prof_bench.json && rm -rf $OUT/prof && head -16 $OUT/kernel_stats.md | cut -c1-140
  bash scripts/pmc_run.sh r03_zb_${w}_pmc --workload $w > $OUT/pmc.log 2>&1; tail -3 $OUT/pmc.log
done
