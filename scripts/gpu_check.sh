#!/bin/bash
# Runs on the MI355X box via gpurun: parity tests, smoke, bench, rocprofv3 kernel stats.
# usage: scripts/gpu_check.sh <tag> [stages...]   stages: tests parity smoke bench prof pmc shard c4
set -u
TAG=${1:-r1}; shift || true
STAGES=${@:-tests smoke bench prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
for st in $STAGES; do
  case $st in
    tests)
      timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
      echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -15 $OUT/pytest_gpu.log ;;
    parity)
      timeout 1500 python -m pytest tests/test_gpu_bench_parity.py -m gpu -q -s -p no:cacheprovider > $OUT/pytest_bench_parity.log 2>&1
      echo "pytest exit $?" >> $OUT/pytest_bench_parity.log; grep -a "parity\|passed\|failed\|Error\|exit" $OUT/pytest_bench_parity.log | tail -25 ;;
    epoch)
      timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -k epoch -p no:cacheprovider > $OUT/pytest_epoch.log 2>&1
      echo "pytest exit $?" >> $OUT/pytest_epoch.log; tail -25 $OUT/pytest_epoch.log ;;
    small)
      bash $ROOT/scripts/sweep_small_batch.sh $TAG ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log; tail -3 $OUT/smoke.log ;;
    bench)
      timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cat $OUT/bench.json; tail -5 $OUT/bench.err ;;
    prof)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err)
      echo "prof exit $?"
      # keep only the summary: the rocpd database is tens of MB and gpurun copies back <= 64 MiB
      db=$(find $OUT/prof -name "*.db" | head -1)
      [ -n "$db" ] && python $ROOT/scripts/summarize_prof.py "$db" $OUT/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline ($TAG)" $OUT/prof_bench.json && rm -rf $OUT/prof && head -30 $OUT/kernel_stats.md ;;
    shard)
      timeout 600 python bench.py --sharded --no-cpu-baseline > $OUT/bench_sharded1.json 2> $OUT/bench_sharded1.err; echo "bench --sharded exit $?"; cat $OUT/bench_sharded1.json; tail -5 $OUT/bench_sharded1.err ;;
    c4)
      timeout 600 python bench.py --workload c4 --steps 8 --warmup 2 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "bench c4 exit $?"; cat $OUT/bench_c4.json; tail -5 $OUT/bench_c4.err ;;
    pmc)
      for c in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && timeout 900 rocprofv3 --pmc $c -d $OUT/pmc_$c -o bench -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err)
        echo "pmc $c exit $?"
      done ;;
  esac
done
rocm-smi --showmeminfo vram 2>/dev/null | head -8 > $OUT/smi.txt
