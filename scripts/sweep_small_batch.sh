#!/bin/bash
# Small-minibatch regime (the reference's own operating points: batch 256 default, 1024 in its tests): the per-minibatch
# launches (epoch_kernel=0) against the persistent epoch kernel (epoch_kernel=1), same box, same process conditions.
# usage: scripts/sweep_small_batch.sh <tag>
TAG=${1:-small}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
run() {  # name users items dim opt batch steps [routes] [extra --set for the persistent route]
  for route in ${8:-0 1}; do
    EXTRA=""
    [ $route = 1 ] && [ -n "${9:-}" ] && EXTRA="--set $9"
    timeout 300 python bench.py --users $2 --items $3 --dim $4 --opt $5 --batch $6 --steps $7 --warmup 50 --no-cpu-baseline \
        --no-probes --no-sharded-check --set epoch_kernel=$route $EXTRA 2> $OUT/err.txt | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read())
except Exception as e:
    print('$1 B=$6 $5 route=$route FAILED', e); sys.exit(0)
r=d['roofline']; k=r['kernels']; o=r['other_ms_per_step']
print(json.dumps({'shape':'$1','users':$2,'items':$3,'dim':$4,'opt':'$5','batch':$6,'steps':d['steps'],'epoch_kernel':$route,'extra':'$EXTRA',
  'us_per_minibatch':d['ms_per_step']*1e3,'M_interactions_per_s':d['value']/1e6,
  'kernel_us_per_minibatch':{'user_pass':k['user_pass']['avg_ms']*1e3 if k['user_pass']['launches'] else 0.0,
    'item_pass':k['item_pass']['avg_ms']*1e3 if k['item_pass']['launches'] else 0.0,'dense_sweep':o['dense_sweep']*1e3,
    'epoch':o['epoch']*1e3,'sample':o['sample']*1e3,'prep':o['prep']*1e3},'loss':d['final_minibatch_loss']}))" | tee -a $OUT/small_batch.jsonl
  done
}
run c1 943 1682 32 adagrad 256 4000
run c1 943 1682 32 adagrad 1024 2000
run c1 943 1682 32 adagrad 2048 1000 "0 1" epoch_max_batch=8192
run c1 943 1682 32 adagrad 4096 1000 "0 1" epoch_max_batch=8192
run c1 943 1682 32 adam_dense 1024 2000 "0 1" epoch_dense_elems=100000000
run c1 943 1682 32 adam_dense 256 4000 "0 1" epoch_dense_elems=100000000
run mid 1000000 100000 64 adagrad 1024 2000
run mid 1000000 100000 64 adagrad 4096 1000 "0 1" epoch_max_batch=8192
run mid 1000000 100000 64 sparse_adam 1024 2000
run c2 10000000 1000000 64 adagrad 1024 2000
run c2 10000000 1000000 64 adagrad 256 4000
cat $OUT/err.txt | tail -5
