#!/bin/bash
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03h; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python scripts/bench_fit_modes.py 33554432 3 > $O/fit_2e25.json 2> $O/fit_2e25.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --user-zipf 1.0 --item-zipf 1.0 --steps 8 --warmup 2 --no-cpu-baseline --no-probes --no-sharded-check --no-fit > $O/prof_zipf.json 2> $O/prof.err)
db=$(find $O/prof -name "*.db" | head -1)
[ -n "$db" ] && python scripts/summarize_prof.py "$db" $O/kernel_stats_zipf.md "rocprofv3 --kernel-trace --stats -- python bench.py --user-zipf 1.0 --item-zipf 1.0 --steps 8 --warmup 2 (r03h)" $O/prof_zipf.json && rm -rf $O/prof && head -16 $O/kernel_stats_zipf.md
for lib in main la main2; do
  if [ $lib = la ]; then export SPOTLIGHT_HIP_LIB=$R/spotlight_amd/csrc/ab/libspotlight_hip_la.so; else unset SPOTLIGHT_HIP_LIB; fi
  python scripts/sweep_engine.py --users 12500000 --items 125000000 --steps 16 --warmup 8 --out $O/c5_$lib.jsonl --configs overlap_prep=0 > $O/c5_$lib.log 2>&1
done
