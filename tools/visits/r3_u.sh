#!/bin/bash
# round 3, visit u: what-if timings of the step (which non-Winograd kernels are on the critical path of the two-stream schedule)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3u
timeout 1200 python tools/gpu_whatif.py 2>&1 | grep -v amdgpu > gpurun_out/r3u/whatif.log
cat gpurun_out/r3u/whatif.log
