#!/bin/bash
# round 5: the ReLU mask inside RCABlock's input-gradient convolution -- backward parity tests, then the training line
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_backward.py -x -q -m gpu > gpurun_out/r5_h_tests.log 2>&1
tail -3 gpurun_out/r5_h_tests.log
timeout 600 python bench.py --mode train --steps 5 --warmup 2 > gpurun_out/r5_h_train.json 2> gpurun_out/r5_h_train.err
tail -c 1500 gpurun_out/r5_h_train.json
