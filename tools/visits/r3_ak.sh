#!/bin/bash
# round 3, visit ak: grouped schedule of conv_wino6_kernel<3> (transform phases of two chunks, pure GEMM iterations): equality + time
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3ak
NS=2,4 timeout 600 python tools/gpu_wino4_timeline.py > gpurun_out/r3ak/timeline.log 2>&1
grep -v "^    [a-su-zA-Z]" gpurun_out/r3ak/timeline.log | head -40
timeout 1200 python -m pytest tests/test_hip_ops.py -m gpu -q -k "wino" 2>&1 | tail -5
