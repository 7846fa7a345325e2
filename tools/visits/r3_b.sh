#!/bin/bash
# round 3, visit b: anatomy of a steady-state iteration (per-wave stamps), bench after the branch-free activation / early U request
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3b
timeout 300 python tools/gpu_wino4_itstamp.py > gpurun_out/r3b/itstamp.log 2>&1
timeout 300 python tools/gpu_wino4_timeline.py > gpurun_out/r3b/timeline.log 2>&1
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -k "conv or wino" 2>&1 | tail -5 > gpurun_out/r3b/tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3b/bench.json 2> gpurun_out/r3b/bench.err
grep -v amdgpu.ids gpurun_out/r3b/itstamp.log
grep -v amdgpu.ids gpurun_out/r3b/timeline.log | head -8
cat gpurun_out/r3b/tests.log
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r3b/bench.json') if l.startswith('{')][-1])
print(d['ms_per_step'], d['value'], d['timed_output_max_abs_vs_eager'], d['roofline']['frac'], d['step_breakdown_ms'])
PY
