#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py -q -k "hot_users or users_that or user_long_gate or closed_loop or at_scale" > $O/pytest_hot.log 2>&1
echo "rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest_hot.log | tail -8
