#!/bin/bash
# round 4: 16-bit heads kernel (conv5x5_c64_h16) with hand-counted LDS waits against the compiler's (libeavsr_f5old.so)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4ad
{
timeout 900 python3 -m pytest tests/test_hip_h16.py -x -q -m gpu 2>&1 | tail -3
for r in 1 2 3; do
  timeout 200 python3 tools/gpu_conv5_h16_ab.py
  EAVSR_LIB_PATH=$PWD/eavsr_amd/lib/libeavsr_f5old.so timeout 200 python3 tools/gpu_conv5_h16_ab.py
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r4ad/log.txt
cat gpurun_out/r4ad/log.txt
