#!/bin/bash
# round 3, visit ai: kernel trace of the bench step -> how much of a step has no CU-owning kernel running
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
mkdir -p gpurun_out/r3ai; rm -rf gpurun_out/r3ai/trace
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3ai/trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile > $R/gpurun_out/r3ai/bench_stdout.log 2>&1
cd $R
head -2 $(find gpurun_out/r3ai/trace -name "*kernel_trace.csv" | head -1) | cut -c1-400
python3 tools/trace_occupancy.py gpurun_out/r3ai/trace > gpurun_out/r3ai/occupancy.log 2>&1
cat gpurun_out/r3ai/occupancy.log
DUMP_AT=150 DUMP_US=900 python3 tools/trace_occupancy.py gpurun_out/r3ai/trace | sed -n "/--- timeline/,\$p" > gpurun_out/r3ai/timeline.log
find gpurun_out/r3ai/trace -name "*.csv" -size +1M -delete
