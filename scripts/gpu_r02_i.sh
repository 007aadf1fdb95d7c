cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02_i; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_engine.py -m gpu -q -k "epoch" -p no:cacheprovider 2>&1 | tail -2
timeout 300 python scripts/bench_c1.py > gpurun_out/r02_i/bench_c1.json 2> gpurun_out/r02_i/bench_c1.err; python -c "
import json; d=json.load(open('gpurun_out/r02_i/bench_c1.json'))
for r in d['runs']: print(r['optimizer'][:30], 'pipelined', r['next_epoch_prepared_while_training'], 'fit_ms %.1f us/mb %.1f' % (r['fit_s']*1e3, r['us_per_minibatch_end_to_end']))"
bash scripts/sweep_small_batch.sh r02_i > /dev/null 2>&1; python -c "
import json
for l in open('gpurun_out/r02_i/small_batch.jsonl'):
    d=json.loads(l); print(d['shape'],d['opt'],d['batch'],'route',d['epoch_kernel'],'us/mb %.1f'%d['us_per_minibatch'])"
