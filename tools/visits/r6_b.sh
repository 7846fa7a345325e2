#!/bin/bash
# round 6: the grouped Winograd kernel's buffer-addressed epilogue -- parity, then old / new in rotation per epilogue form, then the
# headline with the two streams started out of phase (EAVSR_STREAM_SKEW_US)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_model.py -x -q -m gpu -k "wino or conv or rcab or attention or pixel_shuffle or golden or residual" > gpurun_out/r6b/tests.log 2>&1
tail -5 gpurun_out/r6b/tests.log
for form in relu sums rsc res; do
  for rep in 1 2; do
    for lib in hip r5epi; do
      FORM=$form EAVSR_LIB_PATH=$PWD/eavsr_amd/lib/libeavsr_$lib.so timeout 300 python tools/gpu_wino4_ab.py 2>&1 | tail -1
    done
  done
done | tee gpurun_out/r6b/ab.txt
for skew in 0 1500 3000 0 800 2200 4500; do
  EAVSR_STREAM_SKEW_US=$skew timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --also '' > gpurun_out/r6b/bench_skew$skew.json 2> gpurun_out/r6b/bench_skew$skew.err
  python3 - <<PY
import json
d = json.loads(open('gpurun_out/r6b/bench_skew$skew.json').read().strip().splitlines()[-1])
print('skew $skew:', round(d['ms_per_step'], 2), 'ms median', round(d['ms_per_step_median'], 2), d['timed_output_check'].get('bit_identical'), round(d['roofline']['frac'], 4), d['roofline']['canonical_launch']['avg_ms'])
PY
done | tee gpurun_out/r6b/skew.txt
