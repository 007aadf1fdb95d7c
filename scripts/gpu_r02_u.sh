#!/bin/bash
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02_u; mkdir -p $OUT; export TMPDIR=/tmp
for n in 100000000 80000; do
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof$n -o sh -- python $R/scripts/prof_shuffle.py $n > $OUT/prof_$n.txt 2> $OUT/prof_$n.err)
db=$(find $OUT/prof$n -name "*.db" | head -1)
[ -n "$db" ] && python scripts/summarize_prof.py "$db" $OUT/shuffle_kernel_stats_$n.md "rocprofv3 --kernel-trace --stats -- python scripts/prof_shuffle.py $n (two calls)" && rm -rf $OUT/prof$n
cat $OUT/prof_$n.txt; head -34 $OUT/shuffle_kernel_stats_$n.md | cut -c1-110
done
