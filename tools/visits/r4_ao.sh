#!/bin/bash
# round 4: static priority for the computing wave group of conv3x3_c64_h16 (libeavsr_prio{1,2,3}.so) against none (product library)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4ao
{
for r in 1 2; do
for lib in hip prio1 prio2 prio3; do
  echo "== $lib"; EAVSR_LIB_PATH=$PWD/eavsr_amd/lib/libeavsr_$lib.so timeout 300 python3 tools/gpu_h16_sizes.py | tail -5
done
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r4ao/log.txt
cat gpurun_out/r4ao/log.txt
