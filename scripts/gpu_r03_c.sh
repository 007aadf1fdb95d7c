#!/bin/bash
# round 3, call c: overlapped prep as the default + ramped chunks (K dependence), the latency-bound user pass at mid-size minibatches,
# item-pass occupancy 8 variant, hot users (Zipf users: long-run form vs the serial walk), the whole -m gpu suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
S="python scripts/sweep_engine.py"
for kw in "16 4" "20 5" "64 8"; do set -- $kw
  $S --steps $1 --warmup $2 --out $O/c2_K$1.jsonl --configs overlap_prep=0 overlap_prep=1,chunk_ramp=0 overlap_prep=2 > $O/c2_K$1.log 2>&1
done
for b in 65536 8192 2048; do
  $S --batch $b --steps 256 --warmup 32 --out $O/b$b.jsonl --configs user_lat_max_batch=0 overlap_prep=0 overlap_prep=0,user_lat_max_batch=0 \
      item_grid_mult=7 item_grid_mult=6 item_grid_mult=4 > $O/b$b.log 2>&1
done
export SPOTLIGHT_HIP_LIB=$GRAFT_REPO_ROOT/spotlight_amd/csrc/ab/libspotlight_hip_w8.so
$S --out $O/c2_w8.jsonl > $O/c2_w8.log 2>&1
$S --batch 65536 --steps 256 --warmup 32 --out $O/b65536_w8.jsonl --configs item_grid_mult=8 > $O/b65536_w8.log 2>&1
unset SPOTLIGHT_HIP_LIB
for z in 0.8 1.0 1.2; do
  for lib in main nolong; do
    if [ $lib = main ]; then unset SPOTLIGHT_HIP_LIB; else export SPOTLIGHT_HIP_LIB=$GRAFT_REPO_ROOT/spotlight_amd/csrc/ab/libspotlight_hip_$lib.so; fi
    timeout 300 $S --user-zipf $z --steps 16 --warmup 8 --repeat 1 --out $O/uzipf${z}_$lib.jsonl > $O/uzipf${z}_$lib.log 2>&1
  done
done
unset SPOTLIGHT_HIP_LIB
python bench.py --no-cpu-baseline --no-sharded-check > $O/bench.json 2> $O/bench.err
timeout 1200 python -m pytest tests/ -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest_gpu.log | tail -5
