#!/bin/bash
# round 5: input-gradient weights packed in place; PMC blocks of the training step's dominant kernels
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_backward.py tests/test_hip_ops.py -x -q -m gpu > gpurun_out/r5_k_tests.log 2>&1
tail -3 gpurun_out/r5_k_tests.log
timeout 600 python bench.py --mode train --steps 5 --warmup 2 > gpurun_out/r5_k_train.json 2> gpurun_out/r5_k_train.err
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r5_k_train.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['step_breakdown_ms'])
PY
bash tools/gpu_profile_train_kernels.sh > gpurun_out/r5_k_trk.log 2>&1
tail -70 gpurun_out/r5_k_trk.log
