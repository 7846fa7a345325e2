#!/bin/bash
# round 6: the interleaved duty schedule of the F(4x4,3x3) kernel (EAVSR_W4_SCHED=di) -- parity under it, then grp / di in rotation
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6e
EAVSR_W4_SCHED=di timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_model.py -x -q -m gpu -k "wino or conv or rcab or attention or pixel_shuffle or golden or residual or hand_counted" > gpurun_out/r6e/tests_di.log 2>&1
tail -3 gpurun_out/r6e/tests_di.log
for form in sums rsc; do
  for rep in 1 2; do
    for sched in grp di; do
      echo -n "$sched "; EAVSR_W4_SCHED=$sched FORM=$form timeout 300 python tools/gpu_wino4_ab.py 2>&1 | tail -1
    done
  done
done | tee gpurun_out/r6e/ab.txt
for sched in grp di grp di; do
  EAVSR_W4_SCHED=$sched timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --also '' > gpurun_out/r6e/bench_$sched.json 2> gpurun_out/r6e/bench_$sched.err
  python3 - <<PY
import json
d = json.loads(open('gpurun_out/r6e/bench_$sched.json').read().strip().splitlines()[-1])
print('$sched:', round(d['ms_per_step'], 2), 'ms median', round(d['ms_per_step_median'], 2), d['timed_output_check'].get('bit_identical'), round(d['roofline']['frac'], 4), d['roofline']['canonical_launch']['avg_ms'])
PY
done | tee gpurun_out/r6e/bench.txt
