#!/bin/bash
# round 5: the bf16x6 3x3 weight gradient in the training step -- backward parity tests, then the training line
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_backward.py -x -q -m gpu > gpurun_out/r5_j_tests.log 2>&1
tail -3 gpurun_out/r5_j_tests.log
timeout 600 python bench.py --mode train --steps 5 --warmup 2 > gpurun_out/r5_j_train.json 2> gpurun_out/r5_j_train.err
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r5_j_train.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['step_breakdown_ms'])
print({k: v for k, v in d['roofline']['conv_wgrad'].items()})
PY
