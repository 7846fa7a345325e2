#!/bin/bash
# round 4, visit e: il2 defaults (priority on waves 0-3, SLP off): DCNv2 tests, check vs il, ablations + stamps
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4e
timeout 900 python3 -m pytest tests/test_hip_ops.py -q -x -m gpu -k "dcnv2" > gpurun_out/r4e/pytest_dcn.log 2>&1
tail -5 gpurun_out/r4e/pytest_dcn.log
REPS=25 timeout 600 python3 tools/gpu_il2_check.py > gpurun_out/r4e/check.log 2>&1
echo "exit $?" >> gpurun_out/r4e/check.log
tail -9 gpurun_out/r4e/check.log
SIGMA=0.5 timeout 900 python3 tools/gpu_il2_ablate.py > gpurun_out/r4e/ablate.log 2>&1
echo "exit $?" >> gpurun_out/r4e/ablate.log
cat gpurun_out/r4e/ablate.log
