#!/bin/bash
# round 6: the sampler-side DCNv2 backward -- against the column path, the backward parity tests, the training line both ways
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6j
timeout 300 python tools/dbg/dcn_bwd_check.py 2>&1 | tee gpurun_out/r6j/check.txt
timeout 900 python -m pytest tests/test_hip_backward.py -x -q -m gpu > gpurun_out/r6j/tests.log 2>&1
tail -4 gpurun_out/r6j/tests.log
for mode in columns sampler columns sampler; do
  EAVSR_DCN_BWD=$mode timeout 600 python bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r6j/train_$mode.json 2> gpurun_out/r6j/train_$mode.err
  python3 - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r6j/train_$mode.json').read().strip().splitlines()[-1])
    print('$mode:', round(d['ms_per_step'], 2), 'ms', {k: v for k, v in (d.get('step_breakdown_ms') or {}).items() if 'dcn' in k or 'il8' in k})
except Exception as e:
    print('$mode failed', e, open('gpurun_out/r6j/train_$mode.err').read()[-500:])
PY
done | tee gpurun_out/r6j/train.txt
