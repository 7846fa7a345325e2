#!/bin/bash
# round 6: what do the two small launches per RCAB (border sums + ca_scale_pre) cost the STEP?  Ablation: the attention replaced by a
# constant fill (one tiny launch instead of two dependent latency-bound ones; results wrong), per configuration, in rotation.
# (EAVSR_ABLATE_CA_PRE existed in eavsr_amd/ops.py while this measurement was made; removed afterwards: the product has no switch that returns wrong results)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6f
for rep in 1 2; do
 for abl in 0 1; do
  for cfg in 1 2 4; do
    EAVSR_ABLATE_CA_PRE=$abl timeout 900 python bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-profile --also '' > gpurun_out/r6f/b_${cfg}_${abl}.json 2> gpurun_out/r6f/b_${cfg}_${abl}.err
    python3 - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r6f/b_${cfg}_${abl}.json').read().strip().splitlines()[-1])
    print('config $cfg ablate $abl:', round(d['ms_per_step'], 2), 'ms median', round(d['ms_per_step_median'], 2))
except Exception as e:
    print('config $cfg ablate $abl: failed', e, open('gpurun_out/r6f/b_${cfg}_${abl}.err').read()[-400:])
PY
  done
 done
done | tee gpurun_out/r6f/ablate.txt
