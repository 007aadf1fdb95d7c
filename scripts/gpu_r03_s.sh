#!/bin/bash
# refresh of the trace + PMC parts of the final pass with the timed configuration only
TAG=r03_final
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-probes --no-sharded-check --no-fit --no-overlapped > /dev/null 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probes --no-sharded-check --no-fit --no-overlapped > $OUT/prof_bench.json 2> $OUT/prof.err)
db=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$db" ] && python scripts/summarize_prof.py "$db" $OUT/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probes --no-sharded-check --no-fit --no-overlapped ($TAG)" $OUT/prof_bench.json && rm -rf $OUT/prof && head -12 $OUT/kernel_stats.md
bash scripts/pmc_run.sh ${TAG}_pmc --no-probes --no-sharded-check --no-overlapped > $OUT/pmc.log 2>&1; tail -3 $OUT/pmc.log
