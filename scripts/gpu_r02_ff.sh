#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_engine.py -m gpu -q -p no:cacheprovider -k "thousands or epoch_kernel_bit or train_matches_oracle or chunking" 2>&1 | tail -3
LIBS="old new old new" bash scripts/gpu_r02_ee.sh
for z in 0.8 1.0; do python bench.py --item-zipf $z --steps 8 --warmup 2 --no-cpu-baseline --no-probes --no-sharded-check --no-fit 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('zipf $z:', round(d['value']/1e9,3), 'G/s', {k:round(v['avg_ms'],3) for k,v in r['kernels'].items()})"; done
