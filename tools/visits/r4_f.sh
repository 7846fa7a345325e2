#!/bin/bash
# round 4, visit f: ablations of il2 with back-to-back timing (10 launches per sample, clock ramp burnt first)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4f
SIGMA=${SIGMA:-0.5} timeout 900 python3 tools/gpu_il2_ablate.py > gpurun_out/r4f/ablate.log 2>&1
echo "exit $?" >> gpurun_out/r4f/ablate.log
cat gpurun_out/r4f/ablate.log
