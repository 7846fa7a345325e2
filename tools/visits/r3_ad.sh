#!/bin/bash
# round 3, visit ad: conv7x7_f32x6 diagnostics (phase stamps, variants)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3ad
timeout 900 python tools/gpu_conv7_diag.py > gpurun_out/r3ad/diag.log 2>&1
cat gpurun_out/r3ad/diag.log
