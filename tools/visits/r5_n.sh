#!/bin/bash
# round 5: the RCAB's attention before its second convolution (16-bit modes) -- parity tests, then configs[2] / [4] both ways
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_hip_h16.py tests/test_hip_configs.py -x -q -m gpu > gpurun_out/r5_n_tests.log 2>&1
tail -4 gpurun_out/r5_n_tests.log
for cfg in 2 4; do
  for pre in 1 0; do
    EAVSR_RCAB_H16_PRE=$pre timeout 600 python bench.py --config $cfg --steps 4 --warmup 1 --no-cpu-baseline --also '' > gpurun_out/r5_n_c${cfg}_pre$pre.json 2> gpurun_out/r5_n_c${cfg}_pre$pre.err
    python3 - <<PY
import json
d = json.loads(open('gpurun_out/r5_n_c${cfg}_pre$pre.json').read().strip().splitlines()[-1])
print('config $cfg pre $pre:', round(d['ms_per_step'], 2), 'ms', d.get('psnr_vs_fp32', {}).get('psnr_db'), d.get('share_of_step_in_16bit'), d.get('step_breakdown_ms'))
PY
  done
done
