#!/bin/bash
# round 3, call e: hot users with a tile per row group + batched stitch loads; fit() modes; small-batch baselines (PoolNet 256, adaptive hinge)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
S="python scripts/sweep_engine.py"
$S --steps 20 --warmup 5 --out $O/c2_K20.jsonl --configs overlap_prep=0 > $O/c2_K20.log 2>&1
for z in 0.8 1.0 1.2; do
  timeout 300 $S --user-zipf $z --steps 16 --warmup 8 --repeat 1 --out $O/uzipf$z.jsonl > $O/uzipf$z.log 2>&1
done
for z in 1.0 1.2; do
  timeout 300 $S --item-zipf $z --steps 16 --warmup 8 --repeat 1 --out $O/izipf$z.jsonl > $O/izipf$z.log 2>&1
done
timeout 300 $S --user-zipf 1.0 --item-zipf 1.0 --steps 16 --warmup 8 --repeat 1 --out $O/uizipf1.0.jsonl > $O/uizipf1.0.log 2>&1
timeout 300 $S --user-zipf 1.0 --batch 65536 --steps 128 --warmup 32 --repeat 1 --out $O/uzipf1.0_b65536.jsonl > $O/uzipf1.0_b65536.log 2>&1
timeout 600 python scripts/bench_fit_modes.py 33554432 3 > $O/fit_2e25.json 2> $O/fit_2e25.err
timeout 600 python scripts/bench_fit_modes.py 100000000 3 > $O/fit_1e8.json 2> $O/fit_1e8.err
for b in 256 1024; do
  timeout 300 python bench.py --workload c4 --batch $b --seq-len 200 --steps 64 --warmup 16 > $O/c4_b$b.json 2> $O/c4_b$b.err
  timeout 300 python bench.py --workload c4 --batch $b --seq-len 10 --steps 64 --warmup 16 > $O/c4_b${b}_L10.json 2> $O/c4_b${b}_L10.err
done
timeout 300 python scripts/bench_adaptive_small.py > $O/adaptive_small.jsonl 2> $O/adaptive_small.err
echo done
