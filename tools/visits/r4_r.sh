#!/bin/bash
# round 4, visit r: sub-batches per step (2 graphs of 2 clips vs 4 graphs of 1 clip)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4r
for st in 2 4 2 4; do
  timeout 600 python3 bench.py --streams $st --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/r4r/bench_$st.log 2>&1
  echo -n "streams $st: "; tail -1 gpurun_out/r4r/bench_$st.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), d['timed_output_check']['bit_identical'], d.get('degraded'))"
done
