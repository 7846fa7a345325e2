#!/bin/bash
# round 3, visit aa: DCNv2 IL8 kernel, fine MFMA / vector interleave variants against the full kernel (time + equality)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3aa
timeout 900 python tools/gpu_il_ablate.py > gpurun_out/r3aa/ablate.log 2>&1
cat gpurun_out/r3aa/ablate.log
