#!/bin/bash
# round 3, visit aj: timeline of the current conv_wino6_kernel<3> (where the prologue stands after the round's changes)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3aj
NS=2 timeout 600 python tools/gpu_wino4_timeline.py > gpurun_out/r3aj/timeline.log 2>&1
cat gpurun_out/r3aj/timeline.log
