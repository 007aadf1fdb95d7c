#!/bin/bash
# Where the persistent epoch kernel's time goes (C1 shape, minibatch 1024, Adagrad): barrier variants, barriers alone
# (epoch_debug=1), work alone (epoch_debug=2: nobody waits -- results are garbage, only the time means anything).
TAG=${1:-anat}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
one() {  # label, extra bench args...
  local label=$1; shift
  timeout 200 python bench.py --users 943 --items 1682 --dim 32 --opt adagrad --steps 2000 --warmup 50 --no-cpu-baseline --no-probes \
      --no-sharded-check --no-loss-check "$@" 2>> $OUT/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d['roofline']['other_ms_per_step']
print(json.dumps({'case':'$label','us_per_minibatch':d['ms_per_step']*1e3,'epoch_kernel_us':o['epoch']*1e3,'batch':d['config']['global_batch']}))" | tee -a $OUT/anatomy.jsonl
}
for B in 1024 256; do
one "full, one counter + flag"            --batch $B --set epoch_kernel=1
one "full, two-level barrier"             --batch $B --set epoch_kernel=1 --set epoch_barrier=1
one "barriers only, one counter"          --batch $B --set epoch_kernel=1 --set epoch_debug=1
one "barriers only, two-level"            --batch $B --set epoch_kernel=1 --set epoch_debug=1 --set epoch_barrier=1
one "barriers only, no drain"             --batch $B --set epoch_kernel=1 --set epoch_debug=5
one "work only (nobody waits)"            --batch $B --set epoch_kernel=1 --set epoch_debug=2
one "work only, no drain"                 --batch $B --set epoch_kernel=1 --set epoch_debug=6
one "no work, no wait (loop skeleton)"    --batch $B --set epoch_kernel=1 --set epoch_debug=3
done
one "full B=1024 grid 128"                --batch 1024 --set epoch_kernel=1 --set epoch_max_grid=128
one "barriers only B=4096 grid 128"       --batch 4096 --set epoch_kernel=1 --set epoch_debug=1
one "barriers only B=4096 grid 256"       --batch 4096 --set epoch_kernel=1 --set epoch_debug=1 --set epoch_max_grid=256
one "barriers only B=4096 grid 256 2lvl"  --batch 4096 --set epoch_kernel=1 --set epoch_debug=1 --set epoch_max_grid=256 --set epoch_barrier=1
tail -3 $OUT/err.txt
