#!/bin/bash
# round 2, second GPU visit: the rest of the suite, the new DCNv2 IL kernel (tests + timing)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2b
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "il or pair or known_answers" > gpurun_out/r2b/il_tests.log 2>&1; echo "il rc=$?" > gpurun_out/r2b/rc.txt
timeout 300 python tools/gpu_dcn_time.py > gpurun_out/r2b/dcn_time.log 2>&1; echo "time rc=$?" >> gpurun_out/r2b/rc.txt
timeout 600 python -m pytest tests/test_hip_model.py -x -q -m gpu -k "multiadstn" > gpurun_out/r2b/model_tests.log 2>&1; echo "model rc=$?" >> gpurun_out/r2b/rc.txt
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_hip_configs.py > gpurun_out/r2b/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r2b/rc.txt
cat gpurun_out/r2b/rc.txt; cat gpurun_out/r2b/dcn_time.log
tail -n 5 gpurun_out/r2b/il_tests.log gpurun_out/r2b/model_tests.log gpurun_out/r2b/tests.log
