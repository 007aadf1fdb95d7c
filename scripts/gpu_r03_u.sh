#!/bin/bash
# adaptive hinge on the persistent route: bit-identity tests on the GPU + time per minibatch against the launches
mkdir -p gpurun_out/r03_u
timeout 600 python -m pytest tests/test_gpu_engine.py -q -x -k "epoch_kernel" 2>&1 | tail -5 | tee gpurun_out/r03_u/pytest_epoch.txt
timeout 300 python scripts/bench_adaptive_small.py --routes 2>/dev/null | grep '^{' | tee gpurun_out/r03_u/adaptive_routes.jsonl
