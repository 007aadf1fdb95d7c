#!/bin/bash
# round 3, call b: item-pass variants (next-head early loads, next-tile key prefetch) A/B on one box at C2 / B=65536 / C5 shard;
# the new stream probes; the whole -m gpu suite (quota-free assertions)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
for lib in base le pf lepf base2; do
  if [ $lib = base ] || [ $lib = base2 ]; then unset SPOTLIGHT_HIP_LIB; else export SPOTLIGHT_HIP_LIB=$GRAFT_REPO_ROOT/spotlight_amd/csrc/ab/libspotlight_hip_$lib.so; fi
  python scripts/sweep_engine.py --out $O/c2_$lib.jsonl --configs item_grid_mult=8 item_grid_mult=16 item_grid_mult=32 \
      overlap_prep=1,user_grid_mult=6 chunk_interactions=16777216 > $O/c2_$lib.log 2>&1
  python scripts/sweep_engine.py --batch 65536 --steps 256 --warmup 32 --out $O/b65536_$lib.jsonl --configs item_grid_mult=8 item_grid_mult=16 > $O/b65536_$lib.log 2>&1
done
for lib in base lepf; do
  if [ $lib = base ]; then unset SPOTLIGHT_HIP_LIB; else export SPOTLIGHT_HIP_LIB=$GRAFT_REPO_ROOT/spotlight_amd/csrc/ab/libspotlight_hip_$lib.so; fi
  python scripts/sweep_engine.py --users 12500000 --items 125000000 --steps 16 --warmup 8 --out $O/c5_$lib.jsonl --configs item_grid_mult=16 > $O/c5_$lib.log 2>&1
done
unset SPOTLIGHT_HIP_LIB
python bench.py --no-cpu-baseline --no-fit --no-sharded-check > $O/bench.json 2> $O/bench.err
timeout 900 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
