#!/bin/bash
# round 5, visit d: batched weight gradients (uses of a weight as segments of one launch): backward tests, the training line with
# its new breakdown, rocprofv3 census of the training step; il2 tests after the prologue change
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_backward.py -m gpu -x -q > gpurun_out/r5_d_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5_d_tests.log
tail -4 gpurun_out/r5_d_tests.log
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_configs.py tests/test_hip_model.py -m gpu -x -q -k "dcn or il or multiadstn or alignment" > gpurun_out/r5_d_tests2.log 2>&1
tail -2 gpurun_out/r5_d_tests2.log
timeout 600 python bench.py --mode train --steps 5 --warmup 2 > gpurun_out/r5_d_train.json 2> gpurun_out/r5_d_train.err
tail -c 3000 gpurun_out/r5_d_train.json; tail -3 gpurun_out/r5_d_train.err
bash tools/gpu_train_prof.sh > gpurun_out/r5_d_train_prof.log 2>&1
head -40 gpurun_out/r5_d_train_prof.log
