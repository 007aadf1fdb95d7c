#!/bin/bash
# explicit feedback inside the persistent epoch kernel + pipelined explicit fit: GPU tests, then the README-example shape
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/${1:-r02_p}; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_model.py tests/test_sharded.py -m gpu -q -p no:cacheprovider -k "epoch or explicit or pipelined or sharded" 2>&1 | tail -4 | tee $OUT/pytest.txt
timeout 600 python scripts/bench_c1_explicit.py 2>$OUT/c1_explicit.err | tee $OUT/bench_c1_explicit.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for r in d['runs']: print(r['optimizer'][:30], r['persistent_epoch_kernel'], r['next_epoch_prepared_while_training'], round(r['fit_s']*1e3,1),'ms', round(r['us_per_minibatch_end_to_end'],1),'us/mb', round(r['train_rmse_on_uniform_synthetic_data'],4))"
timeout 600 python scripts/bench_c1.py 2>$OUT/c1.err | tee $OUT/bench_c1.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for r in d['runs']: print(r['optimizer'][:30], r['next_epoch_prepared_while_training'], round(r['fit_s']*1e3,1),'ms')"
