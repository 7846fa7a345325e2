#!/bin/bash
# round 3, visit an: do the fp32 convolutions behind a bf16x6 launch run slower (clock hang-over)?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3an
timeout 600 python tools/gpu_clock_hangover.py > gpurun_out/r3an/hangover.log 2>&1
cat gpurun_out/r3an/hangover.log
