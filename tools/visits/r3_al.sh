#!/bin/bash
# round 3, visit al: per-wave stamps of the grouped schedule's iterations 8..11
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3al
timeout 600 python tools/gpu_wino4_itstamp.py > gpurun_out/r3al/itstamp.log 2>&1
cat gpurun_out/r3al/itstamp.log
