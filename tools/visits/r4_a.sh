#!/bin/bash
# round 4, visit a: first run of eavsr_dcnv2_il2_f32 -- agreement with eavsr_dcnv2_il_f32 on the test shapes, timings of both
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4a
timeout 600 python3 tools/gpu_il2_check.py > gpurun_out/r4a/check.log 2>&1
echo "exit $?" >> gpurun_out/r4a/check.log
tail -40 gpurun_out/r4a/check.log
