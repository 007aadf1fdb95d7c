cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02_e
./scripts/micro/coherent_latency > gpurun_out/r02_e/coherent_latency.txt 2>&1; cat gpurun_out/r02_e/coherent_latency.txt
timeout 1500 python -m pytest tests/test_gpu_reference_tests.py -m gpu -q -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/r02_e/reference_tests.txt
for B in 2048; do for route in 0 1; do
python bench.py --users 943 --items 1682 --dim 32 --opt adagrad --batch $B --steps 1000 --warmup 50 --no-cpu-baseline --no-probes --no-sharded-check --set epoch_kernel=$route --set epoch_max_batch=4096 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c1 B=$B route $route us/mb %.1f' % (d['ms_per_step']*1e3))"
python bench.py --users 1000000 --items 100000 --dim 64 --opt adagrad --batch $B --steps 1000 --warmup 50 --no-cpu-baseline --no-probes --no-sharded-check --set epoch_kernel=$route --set epoch_max_batch=4096 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mid B=$B route $route us/mb %.1f' % (d['ms_per_step']*1e3))"
done; done | tee gpurun_out/r02_e/b2048.txt
