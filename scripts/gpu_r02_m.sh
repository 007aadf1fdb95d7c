#!/bin/bash
# fit() end to end at C2 scale: sequential epochs against the pipelined epoch loop (next epoch's shuffle + negatives on the prep lane)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r02_m; mkdir -p $OUT
for lim in 0 1099511627776; do
  SPOTLIGHT_PIPELINE_MAX_DRAWS=$lim timeout 600 python scripts/bench_fit.py 100000000 2>$OUT/fit_$lim.err | tee $OUT/fit_$lim.json
done
