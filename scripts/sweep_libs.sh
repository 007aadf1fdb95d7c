#!/bin/bash
# Same-box A/B of library builds (scripts/ab_build.sh).  usage: scripts/sweep_libs.sh "<bench args>" tag1 tag2 ...
cd ${GRAFT_REPO_ROOT:-.}
ARGS=$1; shift
for tag in "$@"; do
  lib=spotlight_amd/csrc/ab/libspotlight_hip_$tag.so
  [ "$tag" = default ] && lib=spotlight_amd/csrc/libspotlight_hip.so
  SPOTLIGHT_HIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=r.get('kernels',{}); o=r.get('other_ms_per_step', r.get('ms_per_step_by_class', {}))
print('%-10s %-28s %.1f M/s  ms/step %.3f ' % ('$tag', '$ARGS', d['value']/1e6, d['ms_per_step']), {n: round(v['avg_ms'],3) for n,v in k.items()}, {n: round(v,3) for n,v in o.items()})"
done
