#!/bin/bash
# round 4, visit l: does the step answer when the F(4x4,3x3) kernel requests a quarter of its weight pieces (what an in-kernel
# G g G^T expansion would leave), or none?  Whole-library ablation builds (results wrong by construction).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4l
for m in base quarteru noudma base; do
  EAVSR_LIB_PATH=$PWD/eavsr_amd/lib/libeavsr_$m.so timeout 900 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r4l/bench_$m.json 2> gpurun_out/r4l/bench_$m.err
  python3 - "$m" <<'PY'
import json, sys
m = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r4l/bench_{m}.json").read().strip().splitlines()[-1])
    print(m, "step", round(d["ms_per_step"], 2), "ms; conv3x3 64->64 avg", round(d["roofline"]["avg_ms"] * 1e3, 2), "us")
except Exception as e:
    print(m, "failed", e, open(f"gpurun_out/r4l/bench_{m}.err").read()[-300:])
PY
done
