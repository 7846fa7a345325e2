#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2k
timeout 900 python -m pytest tests/test_hip_h16.py -x -q -m gpu > gpurun_out/r2k/h16.log 2>&1; echo "h16 rc=$?" > gpurun_out/r2k/rc.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --backbone-dtype bf16 > gpurun_out/r2k/bench_bf16.json 2> gpurun_out/r2k/bench.err; echo "bench rc=$?" >> gpurun_out/r2k/rc.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --backbone-dtype fp16 > gpurun_out/r2k/bench_fp16.json 2>> gpurun_out/r2k/bench.err
cat gpurun_out/r2k/rc.txt; tail -n 15 gpurun_out/r2k/h16.log
python - <<'PY'
import json
for f in ("gpurun_out/r2k/bench_bf16.json","gpurun_out/r2k/bench_fp16.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["value"],2), round(d["ms_per_step"],2))
        print(d["step_breakdown_ms"])
    except Exception as e: print(f, "ERR", e)
PY
