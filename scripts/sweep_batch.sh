#!/bin/bash
# batch-size sweep of bench.py on one GPU box (same process conditions for A/B)
cd ${GRAFT_REPO_ROOT:-.}
for B in 65536 262144 524288 1048576 2097152 4194304; do
  S=$((16777216 / B)); [ $S -lt 8 ] && S=8; [ $S -gt 64 ] && S=64
  python bench.py --batch $B --steps $S --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('B=%8d steps=%3d  %.1f M int/s  ms/step %.3f  user %.3f item %.3f  sample %.3f prep %.3f' % ($B, d['steps'], d['value']/1e6, d['ms_per_step'], k['user_pass']['avg_ms'], k['item_pass']['avg_ms'], d['roofline']['other_ms_per_step']['sample'], d['roofline']['other_ms_per_step']['prep']))"
done
