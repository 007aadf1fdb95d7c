#!/bin/bash
# banded epoch shuffle: GPU tests, then the shuffle alone and fit() end to end at 1e8 (band on / off), C1 shapes
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/${1:-r02_t}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -p no:cacheprovider -k "shuffle" 2>&1 | tail -3 | tee $OUT/pytest.txt
for band in 1 0; do
  SPOTLIGHT_HIP_OPTIONS=shuffle_band=$band timeout 600 python scripts/bench_fit.py 100000000 2>$OUT/fit_$band.err | tee $OUT/bench_fit_1e8_band$band.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('band $band: shuffle %.1f ms, fit %.1f ms/epoch = %.3f G/s' % (d['device_shuffle_s']*1e3, d['fit_s_per_epoch']*1e3, d['fit_interactions_per_s']/1e9))"
done
timeout 600 python scripts/bench_c1.py 2>$OUT/c1.err | tee $OUT/bench_c1.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for r in d['runs']: print(r['optimizer'][:30], r['next_epoch_prepared_while_training'], round(r['fit_s']*1e3,1),'ms')"
timeout 600 python scripts/bench_c1_explicit.py 2>$OUT/c1e.err | tee $OUT/bench_c1_explicit.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for r in d['runs']: print(r['optimizer'][:30], r['persistent_epoch_kernel'], r['next_epoch_prepared_while_training'], round(r['fit_s']*1e3,1),'ms')"
