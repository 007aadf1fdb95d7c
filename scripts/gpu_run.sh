#!/bin/bash
# The one GPU-side runner (round 4; replaces the one-shot gpu_r0x_*.sh scripts): run on the MI355X box via gpurun.
#   usage: scripts/gpu_run.sh <tag> <stage> [<stage> ...]
# stages (each writes under gpurun_out/<tag>/):
#   tests            pytest -m gpu (whole suite)          tests:<expr>   pytest -m gpu -k <expr>
#   parity           tests/test_gpu_bench_parity.py       smoke          __graft_entry__.smoke()
#   sort             tests/test_gpu_sort.py + scripts/bench_sort.py
#   bench[:<args>]   bench.py <args> (':' separates arguments: bench:--workload:c4:--steps:8)
#   prof[:<args>]    rocprofv3 --kernel-trace --stats of bench.py <args> -> kernel_stats*.md
#   pmc[:<args>]     rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py <args>
#   sweep:<name>:<cfg>[:<cfg>...]   scripts/sweep_engine.py --configs <cfg> ... (cfg = 'opt=v,opt=v'); extra sweep_engine
#                    arguments through SWEEP_ARGS
#   py:<script>[:<args>]   python scripts/<script> <args>
set -u
TAG=${1:-r4}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
for st in "$@"; do
  IFS=':' read -r -a P <<< "$st"
  name=${P[0]}; args=("${P[@]:1}")
  sfx=$(echo "${args[*]:-}" | tr -c 'A-Za-z0-9' '_' | cut -c1-60)
  case $name in
    tests)
      if [ ${#args[@]} -gt 0 ]; then
        timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -k "${args[0]}" > $OUT/pytest_$sfx.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_$sfx.log; tail -8 $OUT/pytest_$sfx.log
      else
        timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -12 $OUT/pytest_gpu.log
      fi ;;
    parity)
      timeout 1500 python -m pytest tests/test_gpu_bench_parity.py -m gpu -q -s -p no:cacheprovider > $OUT/pytest_bench_parity.log 2>&1
      echo "pytest exit $?" >> $OUT/pytest_bench_parity.log; grep -a "parity\|passed\|failed\|Error\|exit" $OUT/pytest_bench_parity.log | tail -25 ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log; tail -3 $OUT/smoke.log ;;
    sort)
      timeout 600 python -m pytest tests/test_gpu_sort.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_sort.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_sort.log; tail -6 $OUT/pytest_sort.log
      timeout 300 python scripts/bench_sort.py ${SORT_ARGS:-} --out $OUT/bench_sort.jsonl 2> $OUT/bench_sort.err | cut -c1-230; tail -3 $OUT/bench_sort.err ;;
    bench)
      timeout 900 python bench.py "${args[@]}" > $OUT/bench_$sfx.json 2> $OUT/bench_$sfx.err; echo "bench ${args[*]:-} exit $?"; cut -c1-1500 $OUT/bench_$sfx.json; tail -3 $OUT/bench_$sfx.err ;;
    prof)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_$sfx -o bench -- python $ROOT/bench.py "${args[@]}" --no-cpu-baseline > $OUT/prof_bench_$sfx.json 2> $OUT/prof_$sfx.err)
      echo "prof exit $?"
      db=$(find $OUT/prof_$sfx -name "*.db" | head -1)
      [ -n "$db" ] && python $ROOT/scripts/summarize_prof.py "$db" $OUT/kernel_stats_$sfx.md "rocprofv3 --kernel-trace --stats -- python bench.py ${args[*]:-} --no-cpu-baseline ($TAG)" $OUT/prof_bench_$sfx.json && rm -rf $OUT/prof_$sfx && head -40 $OUT/kernel_stats_$sfx.md ;;
    pmc)
      for c in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && timeout 900 rocprofv3 --pmc $c -d $OUT/pmc_${c}_$sfx -o bench -- python $ROOT/bench.py "${args[@]}" --no-cpu-baseline > $OUT/pmc_${c}_$sfx.json 2> $OUT/pmc_${c}_$sfx.err)
        echo "pmc $c exit $?"
      done
      python $ROOT/scripts/summarize_pmc.py $OUT > $OUT/pmc_$sfx.md 2>&1; head -30 $OUT/pmc_$sfx.md ;;
    sweep)
      sname=${args[0]}; cfgs=("${args[@]:1}")
      timeout 900 python scripts/sweep_engine.py ${SWEEP_ARGS:-} --out $OUT/sweep_$sname.jsonl --configs "${cfgs[@]}" 2> $OUT/sweep_$sname.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-34s %s  %s' % (d['label'], ['%.4f' % x for x in d.get('ms_per_step_all', [])], {k: round(v, 4) for k, v in d.get('class_ms_per_step', {}).items()}))"
      tail -3 $OUT/sweep_$sname.err ;;
    trace)   # trace:<label>:<script>[:args]  -- rocprofv3 kernel trace of python scripts/<script>
      label=${args[0]}; script=${args[1]}; rest=("${args[@]:2}")
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace_$label -o t -- python $ROOT/scripts/$script "${rest[@]}" > $OUT/trace_$label.out 2> $OUT/trace_$label.err)
      echo "trace $label exit $?"; python scripts/summarize_any.py $OUT/trace_$label > $OUT/trace_$label.md; rm -rf $OUT/trace_$label; head -30 $OUT/trace_$label.md | cut -c1-200 ;;
    fittrace)   # fittrace[:n[:epochs]]  -- kernel trace of the drop-in fit() + per-epoch anatomy
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace -d $OUT/fittrace -o t -- python $ROOT/scripts/trace_fit_epochs.py "${args[@]}" > $OUT/fittrace.out 2> $OUT/fittrace.err)
      echo "fittrace exit $?"; tail -1 $OUT/fittrace.out
      db=$(find $OUT/fittrace -name "*.db" | head -1)
      [ -n "$db" ] && python scripts/fit_epoch_breakdown.py "$db" > $OUT/fit_epoch_breakdown.txt 2>&1 && python scripts/timeline_gaps.py "$db" 100 >> $OUT/fit_epoch_breakdown.txt 2>&1
      rm -rf $OUT/fittrace; head -60 $OUT/fit_epoch_breakdown.txt | cut -c1-200 ;;
    pmcx)    # pmcx:<label>:<counters, space separated>:<script>[:args]  -- one rocprofv3 --pmc pass of python scripts/<script>
      label=${args[0]}; ctrs=${args[1]}; script=${args[2]}; rest=("${args[@]:3}")
      (cd /tmp && timeout 900 rocprofv3 --pmc $ctrs -d $OUT/pmcx_$label -o t -- python $ROOT/scripts/$script "${rest[@]}" > $OUT/pmcx_$label.out 2> $OUT/pmcx_$label.err)
      echo "pmcx $label exit $?"; python scripts/summarize_any.py $OUT/pmcx_$label > $OUT/pmcx_$label.md; rm -rf $OUT/pmcx_$label; head -40 $OUT/pmcx_$label.md | cut -c1-200 ;;
    py)
      script=${args[0]}; rest=("${args[@]:1}")
      timeout 900 python scripts/$script "${rest[@]}" > $OUT/py_$(basename $script .py)_$sfx.log 2>&1; echo "$script exit $?"; tail -25 $OUT/py_$(basename $script .py)_$sfx.log | cut -c1-300 ;;
    *) echo "unknown stage $name" ;;
  esac
done
rocm-smi --showmeminfo vram 2>/dev/null | head -8 > $OUT/smi.txt
