#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3k
timeout 300 python tools/gpu_wino4_timeline.py > gpurun_out/r3k/timeline.log 2>&1
timeout 300 python tools/gpu_wino4_itstamp.py > gpurun_out/r3k/itstamp.log 2>&1
grep -v amdgpu.ids gpurun_out/r3k/timeline.log | grep -v "^    phase\|^    clock"
grep -v amdgpu.ids gpurun_out/r3k/itstamp.log | head -24
