#!/bin/bash
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02_aa; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o b -- python $R/bench.py --workload c4 --batch 256 --steps 64 --warmup 8 > $OUT/bench.json 2> $OUT/err.txt)
db=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$db" ] && python scripts/summarize_prof.py "$db" $OUT/kernel_stats_poolnet_b256.md "rocprofv3 --kernel-trace --stats -- python bench.py --workload c4 --batch 256 --steps 64 --warmup 8" $OUT/bench.json && rm -rf $OUT/prof
head -24 $OUT/kernel_stats_poolnet_b256.md | cut -c1-120
