#!/bin/bash
# round 6, first visit: the whole GPU suite with the new parity tests (DCNv2 out-of-window at the bench shape, gradients at the
# training shape, configs[4] mid size vs the oracle, NaN through the ReLU epilogues, the toolchain guard, stale caches), then the
# headline with the new keys of the line
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6a
(free -g; nproc; hipcc --version | head -2) > gpurun_out/r6a/box.txt 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/r6a/tests.log 2>&1
tail -25 gpurun_out/r6a/tests.log
timeout 900 python bench.py --steps 8 --warmup 3 --also '' > gpurun_out/r6a/bench.json 2> gpurun_out/r6a/bench.err
python3 - <<PY
import json
d = json.loads(open('gpurun_out/r6a/bench.json').read().strip().splitlines()[-1])
print(round(d['ms_per_step'], 2), 'ms', d['timed_output_check'].get('bit_identical'), d['roofline']['frac'], d['config'].get('dcnv2_hbm'), d.get('expected_n1_ms'))
print(d['cpu_baseline'])
print(d['step_breakdown_ms'])
PY
