#!/bin/bash
# round 4, visit c: il2 with inline out-of-window samples, counted vmcnt at the barriers, Ia DMA behind B8: check + ablations
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4c
REPS=15 timeout 600 python3 tools/gpu_il2_check.py > gpurun_out/r4c/check.log 2>&1
echo "exit $?" >> gpurun_out/r4c/check.log
tail -12 gpurun_out/r4c/check.log
timeout 900 python3 tools/gpu_il2_ablate.py > gpurun_out/r4c/ablate.log 2>&1
echo "exit $?" >> gpurun_out/r4c/ablate.log
cat gpurun_out/r4c/ablate.log
