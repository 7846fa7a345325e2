#!/bin/bash
# round 4: full GPU test suite + the default bench (with other_configs) at HEAD
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4aa
timeout 2400 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r4aa/tests.txt
cat gpurun_out/r4aa/tests.txt
timeout 900 python3 bench.py > gpurun_out/r4aa/bench.json 2> gpurun_out/r4aa/bench.err
python3 - <<'PY'
import json
l = json.loads([x for x in open("gpurun_out/r4aa/bench.json") if x.startswith("{")][-1])
print(l["value"], l["ms_per_step"], l["roofline"]["frac"], l["roofline"]["canonical_launch"]["avg_ms"], l["kernels"][0]["avg_ms"], l["kernels"][0]["frac"], l["kernels"][0].get("synthetic_offsets"))
for e in l.get("other_configs", []):
    print({k: e[k] for k in e if k in ("config", "value", "ms_per_step", "share_of_step_in_16bit", "error", "skipped")}, (e.get("roofline") or {}).get("avg_ms"))
PY
