#!/bin/bash
# round 4: conv3x3_c64_h16 with counted vmcnt waits (stores stay in flight) against the build that waits for them
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4w
{
timeout 600 python3 -m pytest tests/test_hip_h16.py -x -q -m gpu 2>&1 | tail -3
echo "== product library"; timeout 300 python3 tools/gpu_h16_sizes.py
echo "== A/B library (libeavsr_waitst.so)"; EAVSR_LIB_PATH=$PWD/eavsr_amd/lib/libeavsr_waitst.so timeout 300 python3 tools/gpu_h16_sizes.py
} > gpurun_out/r4w/log.txt 2>&1
cat gpurun_out/r4w/log.txt
N=16 H=128 W=512 timeout 300 python3 tools/gpu_h16_ablate.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4w/log.txt
