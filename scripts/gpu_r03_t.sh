#!/bin/bash
# first-call penalty: clock / power state or the ids?  (bare ctx default = in line)
mkdir -p gpurun_out/r03_t
for m in inline idle busy w5x2 inline busy idle; do timeout 200 python scripts/probe_first_overlap.py $m 2>/dev/null | tail -1; done | tee gpurun_out/r03_t/first_call.txt
