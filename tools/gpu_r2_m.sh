#!/bin/bash
# round 2, visit m: full GPU suite + configs + smoke + bench (fp32 default and bf16 backbone) after the pair-warp templating,
# the 16-bit kernel report and the wave-specialised DCNv2 entry point
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2m
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_hip_configs.py > gpurun_out/r2m/tests.log 2>&1; echo "tests rc=$?" > gpurun_out/r2m/rc.txt
timeout 1200 python -m pytest tests/test_hip_configs.py -q -m gpu -s > gpurun_out/r2m/configs.log 2>&1; echo "configs rc=$?" >> gpurun_out/r2m/rc.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2m/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r2m/rc.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2m/bench_fp32.json 2> gpurun_out/r2m/bench_fp32.err; echo "bench rc=$?" >> gpurun_out/r2m/rc.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --backbone-dtype bf16 > gpurun_out/r2m/bench_bf16.json 2> gpurun_out/r2m/bench_bf16.err; echo "bench16 rc=$?" >> gpurun_out/r2m/rc.txt
cat gpurun_out/r2m/rc.txt; tail -n 4 gpurun_out/r2m/tests.log; grep "configs\[" gpurun_out/r2m/configs.log; tail -n 3 gpurun_out/r2m/configs.log; tail -3 gpurun_out/r2m/smoke.log
python - <<'PY'
import json
for f in ("fp32", "bf16"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r2m/bench_{f}.json") if l.startswith("{")][-1])
        print(f, d["ms_per_step"], d["value"], d.get("step_breakdown_ms"), d.get("kernels_16bit"), d.get("share_of_step_in_16bit"))
    except Exception as e:
        print(f, "failed", e)
PY
