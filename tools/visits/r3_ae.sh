#!/bin/bash
# round 3, visit ae: conv7x7_f32x6 with the bias in the accumulators, 8-row / 32-channel workgroups for the coarse levels
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3ae
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -k "conv7x7 or conv5x5" 2>&1 | tail -15 > gpurun_out/r3ae/tests.log
timeout 600 python tools/gpu_conv7_time.py > gpurun_out/r3ae/time.log 2>&1

cat gpurun_out/r3ae/tests.log gpurun_out/r3ae/time.log
