#!/bin/bash
# round 5, visit e: fused RCAB-tail backward (one launch behind the plane sums, parameter gradients added in place): backward
# tests, the training line
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_backward.py -m gpu -x -q > gpurun_out/r5_e_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5_e_tests.log
tail -4 gpurun_out/r5_e_tests.log
timeout 600 python bench.py --mode train --steps 5 --warmup 2 > gpurun_out/r5_e_train.json 2> gpurun_out/r5_e_train.err
python - <<'PY'
import json
ln=[l for l in open('gpurun_out/r5_e_train.json') if l.startswith('{')]
d=json.loads(ln[-1]); print(d['value'], d['ms_per_step'], d.get('step_kernel_ms_library')); print(json.dumps(d.get('step_breakdown_ms'))); print(json.dumps(d.get('launches_per_step',{}).get('library_kernels')))
PY
tail -3 gpurun_out/r5_e_train.err
