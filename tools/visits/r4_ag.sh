#!/bin/bash
# round 4: the bench lines at HEAD (default run with other_configs, configs[2] / [4] with their CPU baselines, training)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4ag
timeout 900 python3 bench.py --steps 5 --warmup 2 > gpurun_out/r4ag/bench_line.json 2> gpurun_out/r4ag/bench.err
tail -c 400 gpurun_out/r4ag/bench_line.json
timeout 1200 python3 bench.py --config 2 > gpurun_out/r4ag/bench_line_config2_bf16.json 2> gpurun_out/r4ag/config2.err
timeout 1500 python3 bench.py --config 4 > gpurun_out/r4ag/bench_line_config4_fp16.json 2> gpurun_out/r4ag/config4.err
timeout 900 python3 bench.py --mode train --steps 10 --warmup 3 > gpurun_out/r4ag/bench_line_train.json 2> gpurun_out/r4ag/train.err
ls -la gpurun_out/r4ag
