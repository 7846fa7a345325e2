#!/bin/bash
# round 6: start-up skew between the two streams at configs[2] (16-bit convolutions are 2.5 x shorter: the alignment steps weigh more)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6z
for s in 0 700 1300 2000 0; do
EAVSR_STREAM_SKEW_US=$s timeout 900 python bench.py --config 2 --steps 4 --warmup 1 --no-cpu-baseline --also '' 2>/dev/null | tail -1 | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('config 2 skew $s', round(d['ms_per_step'],2), d.get('timed_output_check',{}).get('bit_identical'))"
done
