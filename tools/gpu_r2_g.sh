#!/bin/bash
# full suite + bench + rocprof after the DCNv2 IL integration
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2g
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_hip_configs.py -x > gpurun_out/r2g/tests.log 2>&1; echo "tests rc=$?" > gpurun_out/r2g/rc.txt
timeout 900 python -m pytest tests/test_hip_configs.py -q -m gpu -x -s -k "config1 or recurrence" > gpurun_out/r2g/configs.log 2>&1; echo "configs rc=$?" >> gpurun_out/r2g/rc.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2g/bench.json 2> gpurun_out/r2g/bench.err; echo "bench rc=$?" >> gpurun_out/r2g/rc.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dcn-mode native > gpurun_out/r2g/bench_native.json 2>> gpurun_out/r2g/bench.err
cat gpurun_out/r2g/rc.txt; tail -n 4 gpurun_out/r2g/tests.log gpurun_out/r2g/configs.log
python - <<'PY'
import json
for f in ("gpurun_out/r2g/bench.json","gpurun_out/r2g/bench_native.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["value"],2), round(d["ms_per_step"],2), [(k["kernel"], round(k["avg_ms"]*1e3,1), round(k["frac"],3)) for k in d["kernels"]])
        print(d["step_breakdown_ms"])
    except Exception as e: print(f, "ERR", e)
PY
