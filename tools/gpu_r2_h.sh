#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2h
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_model.py -x -q -m gpu -k "flow_level or multiadstn or golden" > gpurun_out/r2h/tests.log 2>&1; echo "tests rc=$?" > gpurun_out/r2h/rc.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2h/bench.json 2> gpurun_out/r2h/bench.err; echo "bench rc=$?" >> gpurun_out/r2h/rc.txt
EAVSR_FUSE_LEVEL=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2h/bench_nofuse.json 2>> gpurun_out/r2h/bench.err
cat gpurun_out/r2h/rc.txt; tail -n 4 gpurun_out/r2h/tests.log
python - <<'PY'
import json
for f in ("gpurun_out/r2h/bench.json","gpurun_out/r2h/bench_nofuse.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["value"],2), round(d["ms_per_step"],2), round(d["step_device_ms_instrumented"],1))
        print(d["step_breakdown_ms"])
    except Exception as e: print(f, "ERR", e)
PY
