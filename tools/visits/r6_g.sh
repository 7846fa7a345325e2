#!/bin/bash
# round 6: the border-line sums out of the first convolution's epilogue (desc.border_pieces) -- parity, then the headline both ways
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6g
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_model.py tests/test_hip_configs.py -x -q -m gpu -k "attention_before or wino or rcab or golden or config1_full or hand_counted" > gpurun_out/r6g/tests.log 2>&1
tail -4 gpurun_out/r6g/tests.log
for pcs in 0 1 0 1; do
  EAVSR_RCAB_PRE_PIECES=$pcs timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --also '' > gpurun_out/r6g/bench_$pcs.json 2> gpurun_out/r6g/bench_$pcs.err
  python3 - <<PY
import json
d = json.loads(open('gpurun_out/r6g/bench_$pcs.json').read().strip().splitlines()[-1])
print('pieces $pcs:', round(d['ms_per_step'], 2), 'ms median', round(d['ms_per_step_median'], 2), d['timed_output_check'].get('bit_identical'), round(d['roofline']['frac'], 4), {k: v for k, v in d['step_breakdown_ms'].items() if 'ca_' in k or 'border' in k or '64to64' in k})
PY
done | tee gpurun_out/r6g/bench.txt
