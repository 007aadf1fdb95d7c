#!/bin/bash
# kernel trace of the row-sharded path at world 1 (what the all-to-all's local copy costs beside the engine's kernels)
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r02_o}; mkdir -p $OUT; export TMPDIR=/tmp
for S in 1 4; do
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof$S -o bench -- python $R/bench.py --sharded --slices $S --steps 16 --warmup 4 --shard-chunk 8 --no-cpu-baseline --no-probes > $OUT/prof_bench_s$S.json 2> $OUT/prof$S.err)
db=$(find $OUT/prof$S -name "*.db" | head -1)
[ -n "$db" ] && python scripts/summarize_prof.py "$db" $OUT/kernel_stats_sharded_world1_slices$S.md "rocprofv3 --kernel-trace --stats -- python bench.py --sharded --slices $S --steps 16 --warmup 4 --shard-chunk 8 --no-cpu-baseline --no-probes" $OUT/prof_bench_s$S.json && rm -rf $OUT/prof$S && head -30 $OUT/kernel_stats_sharded_world1_slices$S.md | cut -c1-150
done
