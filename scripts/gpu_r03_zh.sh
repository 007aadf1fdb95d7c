#!/bin/bash
# fit() end to end under the kernel tracer: where the GPU idles between epochs
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03_zh; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $OUT/prof -o fit -- python $R/scripts/bench_fit_modes.py 33554432 4 > $OUT/fit.json 2> $OUT/fit.err)
db=$(find $OUT/prof -name "*.db" | head -1)
python scripts/timeline_gaps.py "$db" 300 k_user_pass | tee $OUT/gaps.txt | head -60
rm -rf $OUT/prof; tail -3 $OUT/fit.json | cut -c1-600
