#!/bin/bash
# round 6: product / lab split -- the GPU suite on the default (product) library, the lab-only tests on the lab library
# (eavsr_amd/lib/libeavsr_lab.so = LAB=1 tools/build_ab_lib.sh lab ""), then the headline
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6d
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6d/tests_default.log 2>&1
tail -3 gpurun_out/r6d/tests_default.log
EAVSR_LIB_PATH=$PWD/eavsr_amd/lib/libeavsr_lab.so timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6d/tests_lab.log 2>&1
tail -3 gpurun_out/r6d/tests_lab.log
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --also '' > gpurun_out/r6d/bench.json 2> gpurun_out/r6d/bench.err
python3 - <<PY
import json
d = json.loads(open('gpurun_out/r6d/bench.json').read().strip().splitlines()[-1])
print(round(d['ms_per_step'], 2), 'ms median', round(d['ms_per_step_median'], 2), d['timed_output_check'].get('bit_identical'), round(d['roofline']['frac'], 4))
PY
