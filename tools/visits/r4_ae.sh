#!/bin/bash
# round 4: the generic 16-bit 3x3 convolution (csrc/conv3_h16.hip, EAVSR_CONV3_16BIT=1 default) against the fp32 Winograd route in
# configs[2] / [4]; tests first
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4ae
{
timeout 1500 python3 -m pytest tests/test_hip_h16.py tests/test_hip_configs.py -x -q -m gpu 2>&1 | tail -5
for cfg in 2 4; do
  for t in 1 0; do
    EAVSR_CONV3_16BIT=$t EAVSR_BREAKDOWN_N=16 timeout 600 python3 bench.py --config $cfg --no-cpu-baseline > gpurun_out/r4ae/c${cfg}_t$t.json
    python3 - <<PY
import json
l = json.loads([x for x in open("gpurun_out/r4ae/c${cfg}_t$t.json") if x.startswith("{")][-1])
print("config $cfg conv3_16=$t", round(l["ms_per_step"], 1), round(l["value"], 2), round(l["share_of_step_in_16bit"], 3), round(l["psnr_vs_fp32"]["psnr_db"], 2), l["timed_output_check"]["bit_identical"])
print("   ", l["step_breakdown_ms"])
PY
  done
done
} > gpurun_out/r4ae/log.txt 2>&1
cat gpurun_out/r4ae/log.txt
