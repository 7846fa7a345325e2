#!/bin/bash
# round 4: A/B of il2 builds in rotation (tools/gpu_il2_ab.py)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4h
timeout 900 python3 tools/gpu_il2_ab.py > gpurun_out/r4h/ab.log 2>&1
echo "exit $?" >> gpurun_out/r4h/ab.log
cat gpurun_out/r4h/ab.log
