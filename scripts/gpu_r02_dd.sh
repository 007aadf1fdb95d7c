#!/bin/bash
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r02_dd}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_engine.py -m gpu -q -p no:cacheprovider -k "thousands or epoch_kernel_bit or train_matches_oracle" 2>&1 | tail -3
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o b -- python $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-probes --no-sharded-check --no-fit > $OUT/bench.json 2> $OUT/err.txt)
db=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$db" ] && python scripts/summarize_prof.py "$db" $OUT/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-probes --no-sharded-check --no-fit" $OUT/bench.json && rm -rf $OUT/prof
head -12 $OUT/kernel_stats.md | cut -c1-110; python -c "
import json; d=json.load(open('$OUT/bench.json')); r=d['roofline']; print('C2 uniform (under rocprof):', round(d['value']/1e9,3), {k:round(v['avg_ms'],4) for k,v in r['kernels'].items()})"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probes --no-sharded-check --no-fit 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('C2 uniform:', round(d['value']/1e9,3), 'G/s', round(d['ms_per_step'],3), 'ms', {k:round(v['avg_ms'],4) for k,v in r['kernels'].items()})"
for z in 0.8 1.0; do python bench.py --item-zipf $z --steps 8 --warmup 2 --no-cpu-baseline --no-probes --no-sharded-check --no-fit 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('zipf $z:', round(d['value']/1e9,3), 'G/s', {k:round(v['avg_ms'],3) for k,v in r['kernels'].items()})"; done
