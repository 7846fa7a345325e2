#!/bin/bash
# round 3, visit ac: step time with the bf16x6 7x7 layers (A/B by EAVSR_CONV7), then the GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3ac
for m in bf16x6 fp32 bf16x6; do
  EAVSR_CONV7=$m timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile > gpurun_out/r3ac/bench_$m.json 2> gpurun_out/r3ac/bench_$m.err
  python - "$m" <<'PY' >> gpurun_out/r3ac/ab.log
import json, sys
m = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r3ac/bench_{m}.json").read().strip().splitlines()[-1])
    print(m, d["ms_per_step"], d["value"], d.get("timed_output_max_abs_vs_eager"))
except Exception as e:
    print(m, "failed", e)
PY
done
cat gpurun_out/r3ac/ab.log
timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r3ac/tests.log
cat gpurun_out/r3ac/tests.log
