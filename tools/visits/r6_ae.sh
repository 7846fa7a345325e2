#!/bin/bash
# round 6: ca_scale_pre's border pieces as one list over sixteen waves -- parity, configs[4] / [2] / [1]
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6ae
timeout 1500 python -m pytest tests/test_hip_h16.py tests/test_hip_ops.py -m gpu -x -q -k "rcab_attention or rcagroup" 2>&1 | tail -3
for c in 4 2 1; do
timeout 900 python bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline --also '' 2>/dev/null | tail -1 | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('config $c', round(d['ms_per_step'],2), d.get('timed_output_check',{}).get('bit_identical'), {k:v for k,v in d.get('step_breakdown_ms',{}).items() if 'ca_' in k})"
done
