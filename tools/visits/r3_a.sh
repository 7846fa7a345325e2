#!/bin/bash
# round 3, visit a: conv_wino6 timeline + schedule variants; baseline bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3a
timeout 600 python tools/gpu_wino4_timeline.py > gpurun_out/r3a/timeline.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
grep -v amdgpu.ids gpurun_out/r3a/timeline.log
tail -c 1500 gpurun_out/r3a/bench.json
