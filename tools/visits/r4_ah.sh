#!/bin/bash
# round 4: ca_scale with 64 loads in flight: CA tests, configs[4] / [2] lines, the default step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4ah
{
timeout 900 python3 -m pytest tests/test_hip_ops.py tests/test_hip_h16.py -x -q -m gpu -k "ca or rca or tail or group" 2>&1 | tail -3
for cfg in 4 2; do
  EAVSR_BREAKDOWN_N=12 timeout 600 python3 bench.py --config $cfg --no-cpu-baseline > gpurun_out/r4ah/c$cfg.json
  python3 - <<PY
import json
l = json.loads([x for x in open("gpurun_out/r4ah/c$cfg.json") if x.startswith("{")][-1])
print("config $cfg", round(l["ms_per_step"], 1), round(l["value"], 2), l["timed_output_check"]["bit_identical"], l["step_breakdown_ms"].get("ca_scale"))
PY
done
timeout 300 python3 bench.py --no-cpu-baseline --no-kernel-profile --also '' --steps 10 | python3 -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', l['ms_per_step'], l['timed_output_check']['bit_identical'])"
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r4ah/log.txt
cat gpurun_out/r4ah/log.txt
