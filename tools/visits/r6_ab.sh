#!/bin/bash
# round 6: the 16-bit RCAB's attention inside its second convolution -- parity, then configs[2] / [4] both ways in rotation
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6ab
timeout 1500 python -m pytest tests/test_hip_h16.py -m gpu -x -q -k "rcab_attention or backbone or h16_model or config" 2>&1 | tail -6
for c in 2 4; do
for s in 1 0 1 0; do
EAVSR_RCAB_H16_ATTN=$s timeout 900 python bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline --also '' 2>/dev/null | tail -1 | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('config $c attn-in-conv $s', round(d['ms_per_step'],2), d.get('timed_output_check',{}).get('bit_identical'), d.get('psnr_vs_fp32',{}).get('psnr_db'), {k:v for k,v in d.get('step_breakdown_ms',{}).items() if 'h16' in k and ('64to64' in k or 'ca_' in k)})"
done; done
