#!/bin/bash
# round 3, visit z: micro-benchmark -- vector instructions in the gap behind an MFMA (own stream / partner wave)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3z
cd tools/ubench && hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_fill mfma_fill.hip 2>/dev/null
timeout 300 /tmp/mfma_fill > "$GRAFT_REPO_ROOT/gpurun_out/r3z/mfma_fill.log" 2>&1
cat "$GRAFT_REPO_ROOT/gpurun_out/r3z/mfma_fill.log"
