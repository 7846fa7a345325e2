#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2i
for n in 2 4; do echo "N=$n"; N=$n timeout 300 python tools/gpu_wino4_ablate.py; done > gpurun_out/r2i/wino_ablate.log 2>&1
cat gpurun_out/r2i/wino_ablate.log
