cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02_h; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for opt in adagrad adam; do
(cd /tmp && C1_OPT=$opt timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$opt -o c1 -- python $R/scripts/trace_c1_fit.py run > $R/gpurun_out/r02_h/run_$opt.txt 2>&1)
f=$(find /tmp/tr_$opt -name "*kernel_trace.csv" | head -1)
echo "$opt: $(grep fit_s gpurun_out/r02_h/run_$opt.txt)"; python scripts/trace_c1_fit.py summarize $f | tee gpurun_out/r02_h/c1_fit_timeline_$opt.json
done
timeout 300 python bench.py --workload c4 --steps 8 --warmup 2 > gpurun_out/r02_h/bench_c4.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r02_h/bench_c4.json')); print('C4', d['value']/1e9, 'G ts/s', d['ms_per_step'], d['roofline']['kernels'], d['roofline']['step_frac_of_peak'])"
timeout 600 python -m pytest tests/test_gpu_engine.py -m gpu -q -k "seq" -p no:cacheprovider 2>&1 | tail -2
