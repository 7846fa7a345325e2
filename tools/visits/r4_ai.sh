#!/bin/bash
# round 4: SPyNet's 7x7 layers with one 16-bit operand plane (EAVSR_CONV7_16BIT=1, default in the 16-bit modes) against the exact bf16x6
# form, configs[2] / [4]; tests first (incl. the fp32 x6 tests: the kernel's template changed)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4ai
{
timeout 1500 python3 -m pytest tests/test_hip_h16.py tests/test_hip_configs.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python3 -m pytest tests/test_hip_ops.py tests/test_hip_model.py -x -q -m gpu -k "x6 or conv7 or spynet or conv5 or heads" 2>&1 | tail -3
for cfg in 2 4; do
  for t in 1 0; do
    EAVSR_CONV7_16BIT=$t EAVSR_BREAKDOWN_N=40 timeout 600 python3 bench.py --config $cfg --no-cpu-baseline > gpurun_out/r4ai/c${cfg}_t$t.json
    python3 - <<PY
import json
l = json.loads([x for x in open("gpurun_out/r4ai/c${cfg}_t$t.json") if x.startswith("{")][-1])
print("config $cfg conv7_16=$t", round(l["ms_per_step"], 1), round(l["value"], 2), round(l["share_of_step_in_16bit"], 3), round(l["psnr_vs_fp32"]["psnr_db"], 2), l["timed_output_check"]["bit_identical"])
print("   ", {k: v for k, v in l["step_breakdown_ms"].items() if "7x7" in k})
PY
  done
done
} > gpurun_out/r4ai/log.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4ai/log.txt
