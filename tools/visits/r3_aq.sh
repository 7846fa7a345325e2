#!/bin/bash
# round 3, visit aq: what-if table of the step at HEAD (which pieces are on the wall clock)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3aq
STEPS=6 timeout 1500 python tools/gpu_whatif.py > gpurun_out/r3aq/whatif.log 2>&1
cat gpurun_out/r3aq/whatif.log
