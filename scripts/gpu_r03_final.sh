#!/bin/bash
# End-of-round verification on one MI355X box (round 3): GPU tests, smoke, the default bench line (with the reference CPU baseline and the
# probes), rocprofv3 kernel stats, PMC traffic of the same command, the other workloads and the batch-size lines.
TAG=${1:-r03_final}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; python -c "
import json; d=json.load(open('$OUT/bench.json')); r=d['roofline']; print('C2', d['value']/1e9, 'G/s', d['ms_per_step'], 'frac', r['frac'], 'step', r['step_frac_of_peak'], r['measured']['variants_GBs'], r['ceiling']['ms_per_step'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline'].get('interactions_per_s_by_threads'))"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probes --no-sharded-check --no-fit --no-overlapped > $OUT/prof_bench.json 2> $OUT/prof.err)
db=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$db" ] && python scripts/summarize_prof.py "$db" $OUT/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probes --no-sharded-check --no-fit --no-overlapped ($TAG)" $OUT/prof_bench.json && rm -rf $OUT/prof && head -12 $OUT/kernel_stats.md
for w in c3 c4 c5; do timeout 600 python bench.py --workload $w --steps 32 --warmup 8 --no-cpu-baseline --no-probes --no-sharded-check > $OUT/bench_$w.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/bench_$w.json')); r=d['roofline']; print('$w', d['value']/1e9, d['unit'], d['ms_per_step'], r.get('step_frac_of_peak'))"; done
for B in 256 1024 65536 1048576; do S=$((16777216 / B)); [ $S -lt 32 ] && S=32; [ $S -gt 2000 ] && S=2000
python bench.py --batch $B --steps $S --warmup 8 --no-cpu-baseline --no-probes --no-sharded-check --no-fit 2>/dev/null | tee $OUT/bench_batch_$B.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C2 tables batch $B: %.1f M interactions/s, %.1f us per minibatch' % (d['value']/1e6, d['ms_per_step']*1e3))"; done
for z in 0.8 1.0 1.2; do python bench.py --item-zipf $z --steps 16 --warmup 8 --no-cpu-baseline --no-probes --no-sharded-check --no-fit 2>/dev/null | tee $OUT/bench_zipf_$z.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('C2 tables, positive items Zipf($z): %.3f G interactions/s, user pass %.3f ms, item pass %.3f ms' % (d['value']/1e9, r['kernels']['user_pass']['avg_ms'], r['kernels']['item_pass']['avg_ms']))"; done
for z in 0.8 1.0 1.2; do python bench.py --user-zipf $z --steps 16 --warmup 8 --no-cpu-baseline --no-probes --no-sharded-check --no-fit 2>/dev/null | tee $OUT/bench_userzipf_$z.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('C2 tables, users Zipf($z): %.3f G interactions/s, user pass %.3f ms, item pass %.3f ms' % (d['value']/1e9, r['kernels']['user_pass']['avg_ms'], r['kernels']['item_pass']['avg_ms']))"; done
python bench.py --user-zipf 1.0 --item-zipf 1.0 --steps 16 --warmup 8 --no-cpu-baseline --no-probes --no-sharded-check --no-fit 2>/dev/null | tee $OUT/bench_bothzipf_1.0.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('C2 tables, users and positive items Zipf(1.0): %.3f G interactions/s, user pass %.3f ms, item pass %.3f ms' % (d['value']/1e9, r['kernels']['user_pass']['avg_ms'], r['kernels']['item_pass']['avg_ms']))"
for opt in sparse_adam; do python bench.py --opt $opt --steps 16 --warmup 4 --no-cpu-baseline --no-probes --no-sharded-check --no-fit 2>/dev/null | tee $OUT/bench_$opt.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('C2 $opt: %.3f G interactions/s, %.4f ms, step frac %.3f' % (d['value']/1e9, d['ms_per_step'], r['step_frac_of_peak']))"; done
timeout 200 python scripts/bench_adaptive_small.py --routes 2>/dev/null | grep '^{' > $OUT/adaptive_small_routes.jsonl; python -c "
import json
for l in open('$OUT/adaptive_small_routes.jsonl'):
    d = json.loads(l); print('adaptive hinge', d['shape'], d['batch'], d['route'], '%.1f us per minibatch' % d['us_per_minibatch'])"
timeout 200 python bench.py --workload c4 --batch 256 --seq-len 10 --items 100000 --steps 400 --warmup 16 2>/dev/null | tee $OUT/bench_poolnet_256x10.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('PoolNet 256 sequences x 10: %.1f us per minibatch' % (d['ms_per_step']*1e3))"
bash scripts/pmc_run.sh ${1:-r03_final}_pmc --no-probes --no-sharded-check --no-overlapped > $GRAFT_REPO_ROOT/gpurun_out/${1:-r03_final}/pmc.log 2>&1; tail -5 $GRAFT_REPO_ROOT/gpurun_out/${1:-r03_final}/pmc.log
