#!/bin/bash
# round 4: the default bench with its `other_configs` leg (configs[2] / [3] / [4] in child processes), wall clock of the whole run
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4v
s=$(date +%s)
timeout 900 python3 bench.py > gpurun_out/r4v/bench.json 2> gpurun_out/r4v/bench.err
echo "exit $? wall $(( $(date +%s) - s )) s" | tee gpurun_out/r4v/wall.txt
python3 - <<'PY'
import json
l = json.loads([x for x in open("gpurun_out/r4v/bench.json") if x.startswith("{")][-1])
print(l["value"], l["ms_per_step"], l["roofline"]["frac"], l["kernels"][0]["avg_ms"], l["kernels"][0]["frac"])
for e in l.get("other_configs", []):
    print({k: e[k] for k in e if k not in ("timed_output_check", "roofline", "psnr_vs_fp32", "workload", "metric")})
    print("   ", e.get("roofline"), e.get("psnr_vs_fp32", {}) and e["psnr_vs_fp32"].get("psnr_db"))
PY
tail -5 gpurun_out/r4v/bench.err
