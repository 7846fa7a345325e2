#!/bin/bash
# round 6: the 5x5 weight gradient in two passes of 50 output tiles instead of 40 + 40 + 20 -- parity, the training line
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6y
timeout 1500 python -m pytest tests/test_hip_backward.py -m gpu -x -q -k "conv2d_backward or wgrad" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_hip_configs.py -m gpu -x -q -k "config3" 2>&1 | tail -3
for c in 1 2; do
timeout 900 python bench.py --mode train --steps 5 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('train', round(d['ms_per_step'],2), d['loss'], {k:v for k,v in d['step_breakdown_ms'].items() if 'wgrad' in k})"
done
