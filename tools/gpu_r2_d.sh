#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2d
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "il or pair or known_answers" > gpurun_out/r2d/il_tests.log 2>&1; echo "il rc=$?" > gpurun_out/r2d/rc.txt
timeout 300 python tools/gpu_il_ablate.py > gpurun_out/r2d/ablate.log 2>&1
timeout 300 python tools/gpu_dcn_time.py > gpurun_out/r2d/dcn_time.log 2>&1
cat gpurun_out/r2d/rc.txt gpurun_out/r2d/ablate.log gpurun_out/r2d/dcn_time.log; tail -n 5 gpurun_out/r2d/il_tests.log
