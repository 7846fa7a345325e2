#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2j
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2j/bench.json 2> gpurun_out/r2j/bench.err
EAVSR_LIB_PATH=$PWD/eavsr_amd/lib/libeavsr_hip_noprio.so timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2j/bench_noprio.json 2>> gpurun_out/r2j/bench.err
python - <<'PY'
import json
for f in ("gpurun_out/r2j/bench.json","gpurun_out/r2j/bench_noprio.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["value"],2), round(d["ms_per_step"],2), round(d["step_device_ms_instrumented"],1), round(d["roofline"]["frac"],3))
        print(d["step_breakdown_ms"])
    except Exception as e: print(f, "ERR", e)
PY
