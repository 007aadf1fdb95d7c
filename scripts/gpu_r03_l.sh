#!/bin/bash
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03l; mkdir -p $O; export TMPDIR=/tmp
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-probes --no-sharded-check --no-fit > /dev/null 2>&1
for ov in 1 0 1 0; do
  python bench.py --workload c4 --steps 40 --warmup 10 --set overlap_prep=$ov 2>/dev/null | tee -a $O/c4_overlap.jsonl | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c4 overlap=$ov: %.4f ms %.3f G ts/s' % (d['ms_per_step'], d['value']/1e9))"
  python bench.py --workload c3 --steps 32 --warmup 8 --set overlap_prep=$ov 2>/dev/null | tee -a $O/c3_overlap.jsonl | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c3 overlap=$ov: %.4f ms %.3f G/s' % (d['ms_per_step'], d['value']/1e9))"
done
timeout 1500 python -m pytest tests/ -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest_gpu.log | tail -8
