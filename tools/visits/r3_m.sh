#!/bin/bash
# round 3, visit m: HIP glue (SPyNet / encoder / tail) -- op test, model + config tests, bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3m
timeout 600 python -m pytest tests/test_hip_ops.py -q -k "glue or resize or pyramid or add" 2>&1 | tail -5 > gpurun_out/r3m/tests.log
timeout 1800 python -m pytest tests/test_hip_model.py tests/test_hip_configs.py -q 2>&1 | tail -8 >> gpurun_out/r3m/tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3m/bench.json 2> gpurun_out/r3m/bench.err
cat gpurun_out/r3m/tests.log
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r3m/bench.json') if l.startswith('{')][-1])
print(d['ms_per_step'], d['value'], d['timed_output_max_abs_vs_eager'], d['roofline']['frac'], d['roofline']['avg_ms'], d['step_breakdown_ms'])
PY
tail -3 gpurun_out/r3m/bench.err
