#!/bin/bash
# round 6: border pieces from the 16-bit first convolution's epilogue -- parity, then configs[2] / [4] / [1] both ways in rotation
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6h
timeout 900 python -m pytest tests/test_hip_h16.py tests/test_hip_ops.py -x -q -m gpu -k "attention_before or rcagroup or h16" > gpurun_out/r6h/tests.log 2>&1
tail -4 gpurun_out/r6h/tests.log
for rep in 1 2; do
 for pcs in 0 1; do
  for cfg in 2 4; do
    EAVSR_RCAB_PRE_PIECES=$pcs timeout 900 python bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-profile --also '' > gpurun_out/r6h/b_${cfg}_${pcs}.json 2> gpurun_out/r6h/b_${cfg}_${pcs}.err
    python3 - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r6h/b_${cfg}_${pcs}.json').read().strip().splitlines()[-1])
    print('config $cfg pieces $pcs:', round(d['ms_per_step'], 2), 'ms median', round(d['ms_per_step_median'], 2), (d.get('psnr_vs_fp32') or {}).get('psnr_db'))
except Exception as e:
    print('config $cfg pieces $pcs: failed', e, open('gpurun_out/r6h/b_${cfg}_${pcs}.err').read()[-600:])
PY
  done
 done
done | tee gpurun_out/r6h/ab.txt
