#!/bin/bash
# round 4, visit s: the round's evidence at HEAD -- rocprofv3 passes (kernel trace of the bench, PMC of the hot kernels), the default
# bench line, the configs[2] / [4] lines with their CPU baselines, the training line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4s
bash tools/gpu_profile.sh r04 > gpurun_out/r4s/profile.log 2>&1
tail -5 gpurun_out/r4s/profile.log | cut -c1-300
cp gpurun_out/prof/summary.txt gpurun_out/r4s/rocprof_summary.txt
cp gpurun_out/prof/bench_kernel_stats.csv gpurun_out/r4s/bench_kernel_stats.csv
cp gpurun_out/prof/traffic_bench.json gpurun_out/r4s/traffic.json
cp gpurun_out/bench_r04.json gpurun_out/r4s/bench_line.json
timeout 1200 python3 bench.py --config 2 > gpurun_out/r4s/bench_line_config2_bf16.json 2> gpurun_out/r4s/config2.err
tail -c 600 gpurun_out/r4s/bench_line_config2_bf16.json
timeout 1500 python3 bench.py --config 4 > gpurun_out/r4s/bench_line_config4_fp16.json 2> gpurun_out/r4s/config4.err
tail -c 600 gpurun_out/r4s/bench_line_config4_fp16.json
timeout 900 python3 bench.py --mode train --steps 10 --warmup 3 > gpurun_out/r4s/bench_line_train.json 2> gpurun_out/r4s/train.err
tail -c 300 gpurun_out/r4s/bench_line_train.json
