#!/bin/bash
# Row-sharded path at world 1 after the block-aligned exchange layout: GPU tests of the path, then bench lines (C2 tables and the C5 shard)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/${1:-r02_n}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_sharded.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/pytest_sharded.txt
for ch in 8 16; do
timeout 300 python bench.py --sharded --steps 16 --warmup 4 --shard-chunk $ch --no-cpu-baseline --no-probes 2>$OUT/sh_c2_$ch.err | tee $OUT/bench_sharded_world1_c2_chunk$ch.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('C2 sharded w1 chunk $ch', d['value']/1e9, d['ms_per_step'], {k:round(v['avg_ms'],3) for k,v in r['kernels'].items()}, r['other_ms_per_step'], r['xgmi']['kernel_ms_per_step'], r['xgmi']['exchange_and_host_ms_per_step'])"
done
timeout 300 python bench.py --sharded --workload c5 --steps 16 --warmup 4 --shard-chunk 16 --no-cpu-baseline --no-probes 2>$OUT/sh_c5.err | tee $OUT/bench_sharded_world1_c5.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('C5 sharded w1', d['value']/1e9, d['ms_per_step'], {k:round(v['avg_ms'],3) for k,v in r['kernels'].items()}, r['other_ms_per_step'], r['xgmi']['kernel_ms_per_step'], r['xgmi']['exchange_and_host_ms_per_step'])"
timeout 300 python bench.py --sharded --slices 4 --steps 16 --warmup 4 --shard-chunk 16 --no-cpu-baseline --no-probes 2>$OUT/sh_c2_s4.err | tee $OUT/bench_sharded_world1_c2_slices4.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('C2 sharded w1 slices 4', d['value']/1e9, d['ms_per_step'], {k:round(v['avg_ms'],3) for k,v in r['kernels'].items()}, r['other_ms_per_step'], r['xgmi']['kernel_ms_per_step'], r['xgmi']['exchange_and_host_ms_per_step'])"
