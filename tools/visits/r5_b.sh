#!/bin/bash
# round 5, visit b: il2 with the rare path on buffer loads + explicit-mode scratch fix: bits vs round 4, tests, rotation timing
# at sigma 0.5 / 4, then the bench step (no CPU baseline, no other configs) for the in-situ DCNv2 figure
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python tools/gpu_il2_ab.py > gpurun_out/r5_b_ab.log 2>&1
SIGMA=4 ROUNDS=2 timeout 600 python tools/gpu_il2_ab.py > gpurun_out/r5_b_ab_s4.log 2>&1
timeout 1200 python -m pytest tests/test_hip_ops.py tests/test_hip_configs.py tests/test_hip_model.py -m gpu -x -q -k "dcn or il or multiadstn or alignment" > gpurun_out/r5_b_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5_b_tests.log
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --also '' > gpurun_out/r5_b_bench.json 2> gpurun_out/r5_b_bench.err
tail -6 gpurun_out/r5_b_ab.log; tail -4 gpurun_out/r5_b_ab_s4.log; tail -3 gpurun_out/r5_b_tests.log
python - <<'PY'
import json
ln=[l for l in open('gpurun_out/r5_b_bench.json') if l.startswith('{')]
d=json.loads(ln[-1]); print(d['value'], d['ms_per_step'], json.dumps(d['roofline'].get('north_star')), d['roofline']['frac'])
PY
