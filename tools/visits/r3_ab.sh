#!/bin/bash
# round 3, visit ab: eavsr_conv7x7_f32x6 -- parity tests, layer timings against the fp32-MFMA kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3ab
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -k "conv7x7" 2>&1 | tail -15 > gpurun_out/r3ab/tests.log
timeout 600 python tools/gpu_conv7_time.py > gpurun_out/r3ab/time.log 2>&1
cat gpurun_out/r3ab/tests.log gpurun_out/r3ab/time.log
