#!/bin/bash
# round 3, call d: hot users after balancing the segments; the settled defaults at several call lengths; the whole -m gpu suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
S="python scripts/sweep_engine.py"
for z in 0.8 1.0 1.2; do
  timeout 300 $S --user-zipf $z --steps 16 --warmup 8 --repeat 1 --out $O/uzipf$z.jsonl > $O/uzipf$z.log 2>&1
done
timeout 300 $S --user-zipf 1.0 --item-zipf 1.0 --steps 16 --warmup 8 --repeat 1 --out $O/uizipf1.0.jsonl > $O/uizipf1.0.log 2>&1
timeout 300 $S --user-zipf 1.0 --batch 65536 --steps 128 --warmup 32 --repeat 1 --out $O/uzipf1.0_b65536.jsonl > $O/uzipf1.0_b65536.log 2>&1
for kw in "20 5" "64 8"; do set -- $kw
  $S --steps $1 --warmup $2 --out $O/c2_K$1.jsonl --configs overlap_prep=0 > $O/c2_K$1.log 2>&1
done
timeout 1500 python -m pytest tests/ -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest_gpu.log | tail -12
