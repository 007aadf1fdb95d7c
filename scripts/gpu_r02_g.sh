cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02_g; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r02_g/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02_g/pytest_gpu.log; tail -12 gpurun_out/r02_g/pytest_gpu.log
timeout 300 python scripts/bench_c1.py > gpurun_out/r02_g/bench_c1.json 2> gpurun_out/r02_g/bench_c1.err; cat gpurun_out/r02_g/bench_c1.json; tail -3 gpurun_out/r02_g/bench_c1.err
timeout 300 python bench.py --sharded --no-cpu-baseline --steps 8 --warmup 2 > gpurun_out/r02_g/bench_sharded1.json 2> gpurun_out/r02_g/sh.err; python -c "
import json; d=json.load(open('gpurun_out/r02_g/bench_sharded1.json')); print('sharded world1 ms/step', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['roofline']['kernels'].items()}, d['roofline']['other_ms_per_step'])"
