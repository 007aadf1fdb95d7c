#!/bin/bash
# item pass with tile partials + k_item_stitch: GPU tests of everything that uses it, C2 uniform (no regression?), Zipf positives
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/${1:-r02_cc}; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/pytest.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probes --no-sharded-check --no-fit 2>/dev/null | tee $OUT/bench_c2.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('C2 uniform:', round(d['value']/1e9,3), 'G/s', round(d['ms_per_step'],3), 'ms', {k:round(v['avg_ms'],4) for k,v in r['kernels'].items()}, r['other_ms_per_step'])"
for z in 0.8 1.0 1.2; do python bench.py --item-zipf $z --steps 8 --warmup 2 --no-cpu-baseline --no-probes --no-sharded-check --no-fit 2>/dev/null | tee $OUT/bench_zipf_$z.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('zipf $z:', round(d['value']/1e9,3), 'G/s', round(d['ms_per_step'],3), 'ms', {k:round(v['avg_ms'],3) for k,v in r['kernels'].items()})"; done
for w in c3 c4 c5; do python bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-probes --no-sharded-check 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w', round(d['value']/1e9,4), round(d['ms_per_step'],4))"; done
