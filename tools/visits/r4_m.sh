#!/bin/bash
# round 4, visit m: DCNv2 tests after the block-placement change, then the round's rocprofv3 passes (kernel trace of the bench +
# PMC passes of the hot kernels, il and il2 side by side)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4m
timeout 900 python3 -m pytest tests/test_hip_ops.py -q -x -m gpu -k "dcnv2" > gpurun_out/r4m/pytest_dcn.log 2>&1
tail -2 gpurun_out/r4m/pytest_dcn.log
bash tools/gpu_profile.sh r04 > gpurun_out/r4m/profile.log 2>&1
tail -70 gpurun_out/r4m/profile.log
cp gpurun_out/prof/summary.txt gpurun_out/r4m/rocprof_summary.txt
cp gpurun_out/prof/bench_kernel_stats.csv gpurun_out/r4m/bench_kernel_stats.csv
cp gpurun_out/prof/traffic_bench.json gpurun_out/r4m/traffic.json
cp gpurun_out/bench_r04.json gpurun_out/r4m/bench_line.json
