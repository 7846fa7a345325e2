#!/bin/bash
# round 4, visit d: schedule variants of il2 (priority, SLP off, non-temporal stores, contiguous-window DMA emulation)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4d
for sg in 0.5 1.5; do
SIGMA=$sg ONLY=full,v_,nofixup timeout 600 python3 tools/gpu_il2_ablate.py > gpurun_out/r4d/ablate_s$sg.log 2>&1
echo "exit $?" >> gpurun_out/r4d/ablate_s$sg.log
echo "--- sigma $sg"; cat gpurun_out/r4d/ablate_s$sg.log
done
