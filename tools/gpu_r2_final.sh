#!/bin/bash
# round-2 evidence: full GPU suite, rocprofv3 stats of the bench command, PMC passes, bench line with CPU baseline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2z
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_hip_configs.py > gpurun_out/r2z/tests.log 2>&1; echo "tests rc=$?" > gpurun_out/r2z/rc.txt
timeout 1200 python -m pytest tests/test_hip_configs.py -q -m gpu -s > gpurun_out/r2z/configs.log 2>&1; echo "configs rc=$?" >> gpurun_out/r2z/rc.txt
bash tools/gpu_profile.sh r02 > gpurun_out/r2z/profile.log 2>&1; echo "profile rc=$?" >> gpurun_out/r2z/rc.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r2z/bench_full.json 2> gpurun_out/r2z/bench_full.err; echo "bench rc=$?" >> gpurun_out/r2z/rc.txt
cat gpurun_out/r2z/rc.txt; tail -n 4 gpurun_out/r2z/tests.log; grep "configs\[" gpurun_out/r2z/configs.log; tail -n 3 gpurun_out/r2z/configs.log
head -40 gpurun_out/prof/summary.txt
tail -c 1500 gpurun_out/r2z/bench_full.json
