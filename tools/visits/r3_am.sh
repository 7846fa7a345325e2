#!/bin/bash
# round 3, visit am: step time, grouped schedule of the 3x3 kernel against the duty-pair schedule (A/B by EAVSR_W4_GRP)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3am
for m in 1 0 1 0; do
  EAVSR_W4_GRP=$m timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile > gpurun_out/r3am/bench_$m.json 2> gpurun_out/r3am/bench_$m.err
  python - "$m" <<'PY' >> gpurun_out/r3am/ab.log
import json, sys
m = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r3am/bench_{m}.json").read().strip().splitlines()[-1])
    print("EAVSR_W4_GRP=" + m, d["ms_per_step"], d["value"], d.get("timed_output_max_abs_vs_eager"))
except Exception as e:
    print(m, "failed", e)
PY
done
cat gpurun_out/r3am/ab.log
