#!/bin/bash
# round 6: the grouped Winograd kernel's epilogue after the store-hazard fix -- bit comparison with the round-5 library, parity
# tests, old / new in rotation, the headline
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6c
timeout 300 python tools/dbg/rsc_diff.py 2>&1 | grep -v "^  " | tee gpurun_out/r6c/rsc_diff.txt
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_model.py -x -q -m gpu -k "wino or conv or rcab or attention or pixel_shuffle or golden or residual" > gpurun_out/r6c/tests.log 2>&1
tail -3 gpurun_out/r6c/tests.log
for form in sums rsc; do
  for rep in 1 2; do
    for lib in hip r5epi; do
      FORM=$form EAVSR_LIB_PATH=$PWD/eavsr_amd/lib/libeavsr_$lib.so timeout 300 python tools/gpu_wino4_ab.py 2>&1 | tail -1
    done
  done
done | tee gpurun_out/r6c/ab.txt
for lib in hip r5epi hip r5epi; do
  EAVSR_LIB_PATH=$PWD/eavsr_amd/lib/libeavsr_$lib.so timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --also '' > gpurun_out/r6c/bench_$lib.json 2> gpurun_out/r6c/bench_$lib.err
  python3 - <<PY
import json
d = json.loads(open('gpurun_out/r6c/bench_$lib.json').read().strip().splitlines()[-1])
print('$lib:', round(d['ms_per_step'], 2), 'ms median', round(d['ms_per_step_median'], 2), d['timed_output_check'].get('bit_identical'), round(d['roofline']['frac'], 4), d['roofline']['canonical_launch']['avg_ms'], d['config'].get('dcnv2_hbm', '')[:30])
PY
done | tee gpurun_out/r6c/bench.txt
