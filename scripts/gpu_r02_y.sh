#!/bin/bash
# rocprofv3 kernel trace of the persistent epoch kernel: C2 tables at minibatch 1024 / 256, and the C1 shape at 1024
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02_y; mkdir -p $OUT; export TMPDIR=/tmp
run() { # tag, bench args...
  local tag=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$tag -o b -- python $R/bench.py "$@" --no-cpu-baseline --no-probes --no-sharded-check --no-fit > $OUT/bench_$tag.json 2> $OUT/$tag.err)
  db=$(find $OUT/prof_$tag -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/summarize_prof.py "$db" $OUT/kernel_stats_epoch_$tag.md "rocprofv3 --kernel-trace --stats -- python bench.py $* --no-cpu-baseline --no-probes --no-sharded-check --no-fit" $OUT/bench_$tag.json && rm -rf $OUT/prof_$tag
  head -16 $OUT/kernel_stats_epoch_$tag.md | cut -c1-120
}
run c2_b1024 --batch 1024 --steps 2000 --warmup 8
run c2_b256 --batch 256 --steps 2000 --warmup 8
run c1_b1024 --users 943 --items 1682 --dim 32 --batch 1024 --steps 2000 --warmup 8
