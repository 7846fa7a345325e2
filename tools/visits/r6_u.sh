#!/bin/bash
# round 6: the plane sums of the RCAB tail's backward out of the next block's input-gradient convolution -- parity, then the
# training line both ways
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6u
timeout 1500 python -m pytest tests/test_hip_backward.py -m gpu -x -q 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_hip_configs.py -m gpu -x -q -k "config3" 2>&1 | tail -4
for c in 1 0 1 0; do
EAVSR_RCAB_CHAIN=$c timeout 900 python bench.py --mode train --steps 5 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('chain $c', round(d['ms_per_step'],2), d['loss'], {k:v for k,v in d['step_breakdown_ms'].items() if k in ('plane_sum','rcab_tail_bwd','conv3x3_64to64_x6s')})"
done
