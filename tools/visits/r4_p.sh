#!/bin/bash
# round 4, visit p: what a balanced DCNv2 launch would take (1 vs 2 tiles per workgroup)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4p
timeout 600 python3 tools/gpu_il2_rounds.py > gpurun_out/r4p/rounds.log 2>&1
cat gpurun_out/r4p/rounds.log | grep -v amdgpu.ids
