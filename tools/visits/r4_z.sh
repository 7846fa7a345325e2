#!/bin/bash
# round 4: Winograd GEMM with hand-counted LDS waits (product library) against the compiler's waits (libeavsr_cread.so):
# tests, the kernel in rotation, then the bench step with each
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4z
{
timeout 900 python3 -m pytest tests/test_hip_ops.py -x -q -m gpu -k "wino or conv" 2>&1 | tail -3
for r in 1 2 3; do
  timeout 200 python3 tools/gpu_wino4_ab.py
  EAVSR_LIB_PATH=$PWD/eavsr_amd/lib/libeavsr_cread.so timeout 200 python3 tools/gpu_wino4_ab.py
done
for r in 1 2; do
  timeout 300 python3 bench.py --no-cpu-baseline --no-kernel-profile --also '' --steps 10 | python3 -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('product', l['ms_per_step'], l['timed_output_check']['bit_identical'])"
  EAVSR_LIB_PATH=$PWD/eavsr_amd/lib/libeavsr_cread.so timeout 300 python3 bench.py --no-cpu-baseline --no-kernel-profile --also '' --steps 10 | python3 -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cread  ', l['ms_per_step'], l['timed_output_check']['bit_identical'])"
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r4z/log.txt
cat gpurun_out/r4z/log.txt
