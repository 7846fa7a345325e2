#!/bin/bash
# round 3, visit g: lock-step kernel with re-paired positions (packed row pass), duty pair exempt from the weight DMA
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3g
timeout 300 python tools/gpu_wino4_timeline.py > gpurun_out/r3g/timeline.log 2>&1
timeout 300 python tools/gpu_wino4_itstamp.py > gpurun_out/r3g/itstamp.log 2>&1
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -k "conv or wino" 2>&1 | tail -5 > gpurun_out/r3g/tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3g/bench.json 2> gpurun_out/r3g/bench.err
grep -v amdgpu.ids gpurun_out/r3g/timeline.log | grep -v "^    \|^  \["
grep -v amdgpu.ids gpurun_out/r3g/itstamp.log | head -24
cat gpurun_out/r3g/tests.log
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r3g/bench.json') if l.startswith('{')][-1])
print(d['ms_per_step'], d['value'], d['timed_output_max_abs_vs_eager'], d['roofline']['frac'], d['roofline']['avg_ms'], d['step_breakdown_ms'])
PY
tail -3 gpurun_out/r3g/bench.err
