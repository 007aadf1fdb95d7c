#!/bin/bash
# same-box A/B of item-pass variants (csrc/ab/libspotlight_hip_<tag>.so; "new" = the built library)
cd $GRAFT_REPO_ROOT
for lib in ${LIBS:-old new}; do
  if [ $lib = new ]; then unset SPOTLIGHT_HIP_LIB; else export SPOTLIGHT_HIP_LIB=$GRAFT_REPO_ROOT/spotlight_amd/csrc/ab/libspotlight_hip_$lib.so; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probes --no-sharded-check --no-fit 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib C2:', round(d['value']/1e9,3), 'G/s', round(d['ms_per_step'],4), 'ms', {k:round(v['avg_ms'],4) for k,v in r['kernels'].items()})"
done
