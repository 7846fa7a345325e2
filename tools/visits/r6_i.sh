#!/bin/bash
# round 6: the whole GPU suite on the default library (after the border-pieces work), then the headline
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6i
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6i/tests_default.log 2>&1
tail -3 gpurun_out/r6i/tests_default.log
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --also '' > gpurun_out/r6i/bench.json 2> gpurun_out/r6i/bench.err
python3 - <<PY
import json
d = json.loads(open('gpurun_out/r6i/bench.json').read().strip().splitlines()[-1])
print(round(d['ms_per_step'], 2), 'ms median', round(d['ms_per_step_median'], 2), d['timed_output_check'].get('bit_identical'), round(d['roofline']['frac'], 4), d['step_breakdown_ms'])
PY
