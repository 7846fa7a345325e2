#!/bin/bash
# round 4: dcnv2_il16 with the fast sigmoid: 16-bit DCN tests, configs[2] / [4] lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4am
{
timeout 1500 python3 -m pytest tests/test_hip_h16.py tests/test_hip_configs.py -x -q -m gpu 2>&1 | tail -2
for cfg in 2 4; do
  EAVSR_BREAKDOWN_N=12 timeout 600 python3 bench.py --config $cfg --no-cpu-baseline | python3 -c "
import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $cfg', round(l['ms_per_step'],1), round(l['psnr_vs_fp32']['psnr_db'],2), l['step_breakdown_ms'].get('dcnv2_il16_heads'), l['step_breakdown_ms'].get('flow_warp_pair'))"
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r4am/log.txt
cat gpurun_out/r4am/log.txt
