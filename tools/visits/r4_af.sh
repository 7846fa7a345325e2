#!/bin/bash
# round 4: conv_last with scalar-loaded weights; the generic 16-bit conv at 2 vs 4 waves per SIMD (libeavsr_lb2.so); configs[2] / [4]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4af
{
timeout 1500 python3 -m pytest tests/test_hip_h16.py -x -q -m gpu 2>&1 | tail -3
for cfg in 2 4; do
  for lib in hip lb2; do
    EAVSR_LIB_PATH=$PWD/eavsr_amd/lib/libeavsr_$lib.so EAVSR_BREAKDOWN_N=40 timeout 600 python3 bench.py --config $cfg --no-cpu-baseline > gpurun_out/r4af/c${cfg}_$lib.json
    python3 - <<PY
import json
l = json.loads([x for x in open("gpurun_out/r4af/c${cfg}_$lib.json") if x.startswith("{")][-1])
print("config $cfg lib $lib", round(l["ms_per_step"], 1), round(l["value"], 2), round(l["share_of_step_in_16bit"], 3), round(l["psnr_vs_fp32"]["psnr_db"], 2), l["timed_output_check"]["bit_identical"])
print("   ", {k: v for k, v in l["step_breakdown_ms"].items() if "h16g" in k or "64to3" in k or "ps2" in k or "wino" in k})
PY
  done
done
} > gpurun_out/r4af/log.txt 2>&1
cat gpurun_out/r4af/log.txt
