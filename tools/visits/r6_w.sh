#!/bin/bash
# round 6: the packed forms of the training step's 3x3 64 -> 64 weights in a handful of launches -- parity, the training line both ways
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6w
timeout 1500 python -m pytest tests/test_hip_backward.py -m gpu -x -q 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_hip_configs.py tests/test_hip_model.py -m gpu -x -q -k "config3 or train or graph" 2>&1 | tail -4
for c in 1 1; do
EAVSR_PREPACK=$c timeout 900 python bench.py --mode train --steps 5 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('prepack $c', round(d['ms_per_step'],2), d['loss'])"
done
