#!/bin/bash
# round 2, visit l: wave-specialised DCNv2 -- parity first, then timing against the symmetric kernel
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_ops.py -q -x -k "il or dcn" 2>&1 | tail -15 > gpurun_out/l_tests.log
timeout 300 python tools/gpu_dcn_time.py > gpurun_out/l_dcn_time.log 2>&1
tail -5 gpurun_out/l_tests.log; cat gpurun_out/l_dcn_time.log
