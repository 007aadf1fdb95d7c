#!/bin/bash
# A/B of engine options on the C4 (PoolNet) workload.  usage: scripts/sweep_c4.sh "<bench args A>" ...
cd ${GRAFT_REPO_ROOT:-.}
for cfg in "$@"; do
  python bench.py --workload c4 --steps 10 --warmup 10 $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels']; o=r['other_ms_per_step']
print('%-50s %.1f M ts/s  ms/step %.3f  seq %.3f item %.3f sample %.3f prep %.3f' % ('$cfg', d['value']/1e6, d['ms_per_step'], k['seq_pass']['avg_ms'], k['item_pass']['avg_ms'], o['sample'], o['prep']))"
done
