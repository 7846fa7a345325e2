#!/bin/bash
# round 5: the RCAB tail of the training forward as one launch; backward / ops parity, then the training line
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_backward.py tests/test_hip_ops.py tests/test_hip_model.py -x -q -m gpu > gpurun_out/r5_l_tests.log 2>&1
tail -3 gpurun_out/r5_l_tests.log
timeout 600 python bench.py --mode train --steps 5 --warmup 2 > gpurun_out/r5_l_train.json 2> gpurun_out/r5_l_train.err
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r5_l_train.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['step_breakdown_ms'])
print(d['launches_per_step']['library_kernels'])
PY
