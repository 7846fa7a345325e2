#!/bin/bash
# round 3, visit ah: step time, 5x5 heads by eavsr_conv_f32x6 against F(2x2,5x5) (A/B by EAVSR_CONV5)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3ah
for m in bf16x6 wino bf16x6 wino; do
  EAVSR_CONV5=$m timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile > gpurun_out/r3ah/bench_$m.json 2> gpurun_out/r3ah/bench_$m.err
  python - "$m" <<'PY' >> gpurun_out/r3ah/ab.log
import json, sys
m = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r3ah/bench_{m}.json").read().strip().splitlines()[-1])
    print(m, d["ms_per_step"], d["value"], d.get("timed_output_max_abs_vs_eager"))
except Exception as e:
    print(m, "failed", e)
PY
done
cat gpurun_out/r3ah/ab.log
