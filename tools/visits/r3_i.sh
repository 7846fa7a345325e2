#!/bin/bash
# round 3, visit i: full GPU suite + bench with the round-3 Winograd kernel defaults
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3i
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3i/tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3i/bench.json 2> gpurun_out/r3i/bench.err
cat gpurun_out/r3i/tests.log
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r3i/bench.json') if l.startswith('{')][-1])
print(d['ms_per_step'], d['value'], d['timed_output_max_abs_vs_eager'], d['roofline']['frac'], d['roofline']['avg_ms'], d['step_breakdown_ms'])
PY
tail -3 gpurun_out/r3i/bench.err
