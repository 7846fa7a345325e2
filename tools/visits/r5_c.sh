#!/bin/bash
# round 5, visit c: il2 after the wait / slot / tf-offset diet: rotation timing vs round 4, il2 tests, then the full profile visit
# (kernel trace of the bench, PMC passes incl. SALU counts, traffic.json, the default bench line with CPU baseline + other configs)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python tools/gpu_il2_ab.py > gpurun_out/r5_c_ab.log 2>&1
timeout 1200 python -m pytest tests/test_hip_ops.py tests/test_hip_configs.py tests/test_hip_model.py -m gpu -x -q -k "dcn or il or multiadstn or alignment" > gpurun_out/r5_c_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5_c_tests.log
tail -5 gpurun_out/r5_c_ab.log; tail -3 gpurun_out/r5_c_tests.log
COMMIT=$1 bash tools/gpu_profile.sh r05 > gpurun_out/r5_c_profile.log 2>&1
tail -5 gpurun_out/r5_c_profile.log
