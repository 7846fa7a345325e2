#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2f
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "il or known_answers" > gpurun_out/r2f/il_tests.log 2>&1; echo "il rc=$?" > gpurun_out/r2f/rc.txt
timeout 300 python tools/gpu_il_ablate.py > gpurun_out/r2f/ablate.log 2>&1
cat gpurun_out/r2f/rc.txt gpurun_out/r2f/ablate.log; tail -n 3 gpurun_out/r2f/il_tests.log
