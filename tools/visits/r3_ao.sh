#!/bin/bash
# round 3, visit ao: how the step responds when the 3x3 kernel loses a part of its work (ablation libraries: results wrong by
# construction, the bench's own timed-vs-eager check still compares like with like) -- is the step bound by that kernel's own time?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3ao
for m in hip nomfma nodma notrans hip; do
  EAVSR_LIB_PATH=$PWD/eavsr_amd/lib/libeavsr_$m.so timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/r3ao/bench_$m.json 2> gpurun_out/r3ao/bench_$m.err
  python - "$m" <<'PY' >> gpurun_out/r3ao/ab.log
import json, sys
m = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r3ao/bench_{m}.json").read().strip().splitlines()[-1])
    print(m, d["ms_per_step"], d["value"], d.get("timed_output_max_abs_vs_eager"))
except Exception as e:
    print(m, "failed", e, open(f"gpurun_out/r3ao/bench_{m}.err").read()[-300:])
PY
done
cat gpurun_out/r3ao/ab.log
