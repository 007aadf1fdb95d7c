#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-probes --no-sharded-check --no-fit > /dev/null 2>&1
for m in inline none inline none; do python scripts/probe_first_overlap.py $m 2>/dev/null | tail -1; done
