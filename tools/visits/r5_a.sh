#!/bin/bash
# round 5, visit a: the buffer-addressed il2 kernel against round 4's (libil2_r4.so): bits, tests, rotation timing
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python tools/gpu_il2_ab.py > gpurun_out/r5_a_ab.log 2>&1
echo "ab rc=$?" >> gpurun_out/r5_a_ab.log
timeout 1200 python -m pytest tests/test_hip_ops.py tests/test_hip_configs.py -m gpu -x -q -k "dcn or il or multiadstn or alignment" > gpurun_out/r5_a_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5_a_tests.log
SIGMA=2 ROUNDS=2 timeout 600 python tools/gpu_il2_ab.py > gpurun_out/r5_a_ab_s2.log 2>&1
SIGMA=4 ROUNDS=2 timeout 600 python tools/gpu_il2_ab.py > gpurun_out/r5_a_ab_s4.log 2>&1
tail -8 gpurun_out/r5_a_ab.log; tail -5 gpurun_out/r5_a_tests.log; tail -4 gpurun_out/r5_a_ab_s2.log; tail -4 gpurun_out/r5_a_ab_s4.log
