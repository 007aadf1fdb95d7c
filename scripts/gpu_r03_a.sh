#!/bin/bash
# round 3, call a: overlap of the chunk prep with the passes under a chip partition (sweep), new bench-size parity tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
python scripts/sweep_engine.py --out $O/sweep.jsonl --configs \
  overlap_prep=1 overlap_prep=2 \
  user_grid_mult=7 user_grid_mult=6 user_grid_mult=5 \
  overlap_prep=1,user_grid_mult=7 overlap_prep=1,user_grid_mult=6 overlap_prep=1,user_grid_mult=5 overlap_prep=1,user_grid_mult=4 \
  overlap_prep=2,user_grid_mult=6 \
  overlap_prep=1,prep_priority=1 overlap_prep=1,prep_priority=1,user_grid_mult=6 overlap_prep=2,prep_priority=1,user_grid_mult=6 \
  overlap_prep=1,prep_cus=16 overlap_prep=1,prep_cus=32 overlap_prep=1,prep_cus=48 overlap_prep=1,prep_cus=64 \
  overlap_prep=2,prep_cus=8 overlap_prep=2,prep_cus=16 overlap_prep=2,prep_cus=32 \
  > $O/sweep.log 2>&1
echo "sweep rc=$?"
tail -3 $O/sweep.log
timeout 900 python -m pytest tests/test_gpu_bench_parity.py -x -q -s -k "saturated or sparse_adam or small_and_mid or sharded_world1" > $O/parity.log 2>&1
echo "parity rc=$?"
tail -5 $O/parity.log
