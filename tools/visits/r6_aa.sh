#!/bin/bash
# round 6: the batched weight-gradient launches on a side stream (they fill the CUs the crop-sized convolution chain leaves idle) --
# parity, graphed-vs-eager, then the training line both ways
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6aa
timeout 1500 python -m pytest tests/test_hip_backward.py -m gpu -x -q 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_hip_configs.py tests/test_hip_model.py -m gpu -x -q -k "config3 or train or graph" 2>&1 | tail -4
for c in 1 0 1 0; do
EAVSR_WGRAD_STREAM=$c timeout 900 python bench.py --mode train --steps 5 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('side stream $c', round(d['ms_per_step'],2), d['loss'])"
done
