#!/bin/bash
# round 4: conv_x6 patch loads with scalar bases against the previous library: x6 tests, heads / 7x7 times in the default line and configs[4]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4an
{
timeout 900 python3 -m pytest tests/test_hip_ops.py tests/test_hip_h16.py -x -q -m gpu -k "x6 or conv7 or conv5 or heads or h16x1" 2>&1 | tail -2
for lib in hip prev hip prev; do
  EAVSR_LIB_PATH=$PWD/eavsr_amd/lib/libeavsr_$lib.so EAVSR_BREAKDOWN_N=40 timeout 300 python3 bench.py --no-cpu-baseline --also '' --steps 8 | python3 -c "
import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=l['step_breakdown_ms']; print('$lib default', round(l['ms_per_step'],2), {k: b[k] for k in b if 'x6' in k}, l['timed_output_check']['bit_identical'])"
done
for lib in hip prev; do
  EAVSR_LIB_PATH=$PWD/eavsr_amd/lib/libeavsr_$lib.so EAVSR_BREAKDOWN_N=40 timeout 300 python3 bench.py --config 4 --no-cpu-baseline | python3 -c "
import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=l['step_breakdown_ms']; print('$lib config4', round(l['ms_per_step'],2), {k: b[k] for k in b if '7x7' in k})"
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r4an/log.txt
cat gpurun_out/r4an/log.txt
