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
# what the passes wait for: requests the L2 sends to the memory side, how many are outstanding on average
# (LEVEL / REQ = cycles a request stays outstanding), and the cycles those queues refuse new ones
if [ -n "${PMC_EA:-}" ]; then
  run ea_rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
  run ea_wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum
  # (a TA_* group -- TA_ADDR_STALLED_BY_TC_CYCLES_sum, TA_DATA_STALLED_BY_TC_CYCLES_sum, TA_TA_BUSY_sum -- never returned on
  # this pool: rocprofv3 sat until the time-out, twice 600 s of box time in round 4.  Not collected.)
fi
python $ROOT/scripts/summarize_pmc.py $OUT $OUT/pmc_summary "rocprofv3 --pmc passes over python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fit ${BARGS[*]} ($TAG)" > /dev/null
for g in fetch write sq tcc utcl1 tcpreq ea_rd ea_wr; do rm -rf $OUT/$g; done
ls $OUT; cat $OUT/pmc_summary.md | head -60
