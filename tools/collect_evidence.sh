#!/bin/bash
# copy the outputs of an evidence visit (tools/visits/r6_n.sh) from the scratch gpurun_out/ into the tracked profiles/
# usage: tools/collect_evidence.sh r06 r6_n
set -e
cd "$(dirname "$0")/.."
TAG=$1; V=$2
cp gpurun_out/prof/summary.txt profiles/${TAG}_rocprof_summary.txt
cp gpurun_out/prof/bench_kernel_stats.csv profiles/${TAG}_bench_kernel_stats.csv
tail -1 gpurun_out/bench_${TAG}.json > profiles/${TAG}_bench_line.json
tail -1 gpurun_out/${V}_train.json > profiles/${TAG}_bench_line_train.json
cp gpurun_out/${V}_train_prof.log profiles/${TAG}_rocprof_summary_train.txt
cp gpurun_out/${V}_h16.log profiles/${TAG}_rocprof_summary_h16.txt
cp gpurun_out/prof/traffic_bench.json profiles/${TAG}_traffic.json
cp gpurun_out/prof/traffic_bench.json profiles/traffic.json
ls -la profiles/${TAG}_*
