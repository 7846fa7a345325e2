#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3at
timeout 600 python tools/gpu_stream_skew.py > gpurun_out/r3at/skew.log 2>&1
cat gpurun_out/r3at/skew.log
