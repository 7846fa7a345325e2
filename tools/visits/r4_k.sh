#!/bin/bash
# round 4, visit k: F(4x4,3x3) kernel with the grid capped (EAVSR_W4_MAXGRID): step time, two streams
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4k
for mg in 256 128 115 160 77; do
  EAVSR_W4_MAXGRID=$mg timeout 600 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/r4k/bench_$mg.log 2>&1
  echo -n "maxgrid $mg: "; tail -1 gpurun_out/r4k/bench_$mg.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['timed_output_check']['bit_identical'])"
done
