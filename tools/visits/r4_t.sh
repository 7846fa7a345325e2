#!/bin/bash
# round 4, visit t: the whole library with the SLP vectorizer off (A/B in the step)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4t
for m in ${LIBS:-base noslp base noslp}; do
  EAVSR_LIB_PATH=$PWD/eavsr_amd/lib/libeavsr_$m.so timeout 900 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r4t/bench_$m.json 2> gpurun_out/r4t/bench_$m.err
  python3 - "$m" <<'PY'
import json, sys
m = sys.argv[1]
d = json.loads(open(f"gpurun_out/r4t/bench_{m}.json").read().strip().splitlines()[-1])
k = {e['kernel']: e for e in d['kernels']}
print(m, "step", round(d["ms_per_step"], 2), "ms; conv3x3 64->64", round(d["roofline"]["avg_ms"] * 1e3, 2), "us; heads 5x5", round(k['conv5x5_64to120_x6']['avg_ms']*1e3,1), "; dcn", round(k['dcnv2_il_heads']['avg_ms']*1e3,1), "; breakdown", {a: b for a, b in list(d['step_breakdown_ms'].items())[:6]})
PY
done
