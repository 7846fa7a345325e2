#!/bin/bash
# round 4, visit b: timing ablations + per-phase stamps of eavsr_dcnv2_il2_f32 (tools/build_il2_diag.sh variants)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4b
timeout 900 python3 tools/gpu_il2_ablate.py > gpurun_out/r4b/ablate.log 2>&1
echo "exit $?" >> gpurun_out/r4b/ablate.log
cat gpurun_out/r4b/ablate.log
