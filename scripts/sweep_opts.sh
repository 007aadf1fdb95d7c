#!/bin/bash
# A/B of engine tuning options on one GPU box.  usage: scripts/sweep_opts.sh "<bench args A>" "<bench args B>" ...
cd ${GRAFT_REPO_ROOT:-.}
for cfg in "$@"; do
  python bench.py --no-cpu-baseline $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; o=d['roofline']['other_ms_per_step']
print('%-60s %.1f M/s  ms/step %.3f (timers %.3f)  user %.3f item %.3f sample %.3f prep %.3f' % ('$cfg', d['value']/1e6, d['ms_per_step'], d['ms_per_step_with_kernel_timers'], k['user_pass']['avg_ms'], k['item_pass']['avg_ms'], o['sample'], o['prep']))"
done
