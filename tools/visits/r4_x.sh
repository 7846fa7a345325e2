#!/bin/bash
# round 4: conv3x3_c64_h16 ablations + stamps at a steady-state size (16 tiles per CU) and at the bench shapes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4x
{
N=16 H=128 W=512 timeout 300 python3 tools/gpu_h16_ablate.py
N=2,4 timeout 300 python3 tools/gpu_h16_ablate.py
} > gpurun_out/r4x/log.txt 2>&1
cat gpurun_out/r4x/log.txt
