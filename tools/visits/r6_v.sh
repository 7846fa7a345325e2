#!/bin/bash
# round 6: ca_tail (training form) with its requests in front of the barriers -- parity, the training line
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6v
timeout 1500 python -m pytest tests/test_hip_backward.py tests/test_hip_ops.py -m gpu -x -q -k "rcab or ca_tail or tail or attention" 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_hip_configs.py tests/test_hip_model.py -m gpu -x -q -k "config3 or train" 2>&1 | tail -4
for c in 1 2; do
timeout 900 python bench.py --mode train --steps 5 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('train', round(d['ms_per_step'],2), d['loss'], {k:v for k,v in d['step_breakdown_ms'].items() if k in ('ca_tail','rcab_tail_bwd','conv3x3_64to64_x6s')})"
done
