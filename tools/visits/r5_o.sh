#!/bin/bash
# round 5: the fp32 RCAB's attention before its second convolution (EAVSR_RCAB_PRE) -- parity, then the headline both ways
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "attention_before or wino" > gpurun_out/r5_o_tests.log 2>&1
tail -3 gpurun_out/r5_o_tests.log
for pre in 0 1 0 1; do
  EAVSR_RCAB_PRE=$pre timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --also '' > gpurun_out/r5_o_pre$pre.json 2> gpurun_out/r5_o_pre$pre.err
  python3 - <<PY
import json
d = json.loads(open('gpurun_out/r5_o_pre$pre.json').read().strip().splitlines()[-1])
print('pre $pre:', round(d['ms_per_step'], 2), 'ms', d['timed_output_check'].get('bit_identical'), d['roofline']['north_star']['frac'], d['step_breakdown_ms'] if 'step_breakdown_ms' in d else '')
PY
done
