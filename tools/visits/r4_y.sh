#!/bin/bash
# round 4: 16-bit tests + configs[2] / [4] bench lines after the conv3x3_c64_h16 changes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4y
{
timeout 1200 python3 -m pytest tests/test_hip_h16.py tests/test_hip_configs.py -x -q -m gpu 2>&1 | tail -3
EAVSR_BREAKDOWN_N=30 timeout 600 python3 bench.py --config 2 --no-cpu-baseline > gpurun_out/r4y/c2.json
EAVSR_BREAKDOWN_N=30 timeout 600 python3 bench.py --config 4 --no-cpu-baseline > gpurun_out/r4y/c4.json
python3 - <<'PY'
import json
for f in ("c2", "c4"):
    l = json.loads([x for x in open(f"gpurun_out/r4y/{f}.json") if x.startswith("{")][-1])
    print(f, l["value"], l["ms_per_step"], l["share_of_step_in_16bit"], l["roofline"]["avg_ms"], l["roofline"]["frac"], l["roofline"]["hbm"]["frac"], l["psnr_vs_fp32"]["psnr_db"])
    print(l["step_breakdown_ms"])
PY
} > gpurun_out/r4y/log.txt 2>&1
cat gpurun_out/r4y/log.txt
