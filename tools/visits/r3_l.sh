#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3l
timeout 300 python tools/gpu_wino4_timeline.py > gpurun_out/r3l/timeline.log 2>&1
grep -v amdgpu.ids gpurun_out/r3l/timeline.log | grep -v "^    \|^  \["
