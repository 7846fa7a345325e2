#!/bin/bash
# round 2, first GPU visit: new parity tests, the full suite, the bench line with the corrected fields
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2a
nproc > gpurun_out/r2a/nproc.txt; lscpu | head -20 >> gpurun_out/r2a/nproc.txt
timeout 1500 python -m pytest tests/test_hip_configs.py -x -q -m gpu -s > gpurun_out/r2a/configs.log 2>&1; echo "configs rc=$?" >> gpurun_out/r2a/rc.txt
timeout 1200 python -m pytest tests -x -q -m gpu --deselect tests/test_hip_configs.py > gpurun_out/r2a/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r2a/rc.txt
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err; echo "bench rc=$?" >> gpurun_out/r2a/rc.txt
timeout 300 python bench.py --steps 5 --warmup 2 --no-kernel-profile --cpu-threads 32 --cpu-runs 1 > gpurun_out/r2a/bench_cpu32.json 2>> gpurun_out/r2a/bench.err
cat gpurun_out/r2a/rc.txt
tail -5 gpurun_out/r2a/configs.log gpurun_out/r2a/tests.log
