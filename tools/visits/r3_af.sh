#!/bin/bash
# round 3, visit af: step time at HEAD (twice), GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3af
for i in 1 2; do
  timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile > gpurun_out/r3af/bench_$i.json 2> gpurun_out/r3af/bench_$i.err
  python - "$i" <<'PY' >> gpurun_out/r3af/ab.log
import json, sys
m = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r3af/bench_{m}.json").read().strip().splitlines()[-1])
    print(m, d["ms_per_step"], d["value"], d.get("timed_output_max_abs_vs_eager"))
except Exception as e:
    print(m, "failed", e)
PY
done
cat gpurun_out/r3af/ab.log
timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r3af/tests.log
cat gpurun_out/r3af/tests.log
