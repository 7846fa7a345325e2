#!/bin/bash
# round 4: flow_warp_pair's NCHW half with 4 channels (16 loads) in flight against 1 (libeavsr_wu1.so): kernel time in the default line,
# configs[2] step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4aj
{
timeout 600 python3 -m pytest tests/test_hip_ops.py -x -q -m gpu -k "warp" 2>&1 | tail -2
for lib in hip wu1 hip wu1; do
  EAVSR_LIB_PATH=$PWD/eavsr_amd/lib/libeavsr_$lib.so timeout 300 python3 bench.py --no-cpu-baseline --also '' --steps 8 | python3 -c "
import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=[e for e in l['kernels'] if e['kernel']=='flow_warp_pair'][0]; print('$lib default', round(l['ms_per_step'],2), 'warp_pair us', round(k['avg_ms']*1e3,1), l['timed_output_check']['bit_identical'])"
  EAVSR_LIB_PATH=$PWD/eavsr_amd/lib/libeavsr_$lib.so timeout 300 python3 bench.py --config 2 --no-cpu-baseline | python3 -c "
import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib config2', round(l['ms_per_step'],2), l['step_breakdown_ms'].get('flow_warp_pair'))"
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r4aj/log.txt
cat gpurun_out/r4aj/log.txt
