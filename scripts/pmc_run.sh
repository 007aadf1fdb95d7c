#!/bin/bash
# Runs on the MI355X box via gpurun: rocprofv3 PMC passes over bench.py (one counter group per pass, no
# tracing flags -- MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots").   usage: scripts/pmc_run.sh <tag> [bench args]
set -u
TAG=${1:-pmc}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() {  # name, counters...
  local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" -d $OUT/$name -o pmc -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fit "${BARGS[@]}" > $OUT/$name.json 2> $OUT/$name.err
  echo "pmc $name exit $?"
}
BARGS=("$@")
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
# address translation (the 1B-item regime: tables far larger than the TLBs' reach); counter names as `rocprofv3 -L` lists them on gfx950
if [ -n "${PMC_TLB:-}" ]; then
  rocprofv3 -L 2>/dev/null | grep -o -i "[A-Z0-9_]*UTCL[A-Z0-9_]*" | sort -u > $OUT/utcl_counters_available.txt
  run utcl1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum
  run tcpreq TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
fi
python $ROOT/scripts/summarize_pmc.py $OUT $OUT/pmc_summary "rocprofv3 --pmc passes over python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fit ${BARGS[*]} ($TAG)" > /dev/null
for g in fetch write sq tcc utcl1 tcpreq; do rm -rf $OUT/$g; done
ls $OUT; cat $OUT/pmc_summary.md | head -60
