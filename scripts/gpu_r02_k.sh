cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02_k
timeout 600 python -m pytest tests/test_gpu_engine.py -m gpu -q -k "epoch" -p no:cacheprovider 2>&1 | tail -2
for B in 256 1024; do for route in 0 1; do
python bench.py --users 943 --items 1682 --dim 32 --opt adam_dense --batch $B --steps 2000 --warmup 50 --no-cpu-baseline --no-probes --no-sharded-check --set epoch_kernel=$route 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c1 adam_dense B=$B route $route us/mb %.1f' % (d['ms_per_step']*1e3))"
done; done | tee gpurun_out/r02_k/dense.txt
python scripts/bench_c1.py 2>/dev/null > gpurun_out/r02_k/bench_c1.json; python -c "
import json; d=json.load(open('gpurun_out/r02_k/bench_c1.json'))
for r in d['runs']: print(r['optimizer'][:30], 'pipelined', r['next_epoch_prepared_while_training'], 'fit_ms %.1f us/mb %.1f' % (r['fit_s']*1e3, r['us_per_minibatch_end_to_end']))"
