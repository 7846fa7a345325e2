#!/bin/bash
# round 2: numbers for DESIGN.md besides the headline -- the other BASELINE configs, the training step, a long run of the bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2num
timeout 900 python tools/gpu_configs.py > gpurun_out/r2num/configs.log 2>&1
timeout 600 python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2num/train.json 2> gpurun_out/r2num/train.err
timeout 600 python bench.py --mode train --graph --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2num/train_graph.json 2> gpurun_out/r2num/train_graph.err
timeout 900 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/r2num/long.json 2> gpurun_out/r2num/long.err
grep -v amdgpu gpurun_out/r2num/configs.log
python - <<'PY'
import json
for f in ("train", "train_graph", "long"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r2num/{f}.json") if l.startswith("{")][-1])
        print(f, d["metric"], d["value"], d["unit"], d["ms_per_step"], d.get("ms_per_step_median"), d.get("loss"))
    except Exception as e:
        print(f, "failed", e)
PY
