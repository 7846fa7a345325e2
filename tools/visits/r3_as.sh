#!/bin/bash
# round 3, visit as: bench.py --gpus 2 on a one-GPU box (two ranks share the device; gloo for the host scalars) at HEAD
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3as
bash tools/gpu_two_rank_bench.sh > gpurun_out/r3as/two_rank.log 2>&1
tail -c 1500 gpurun_out/r3as/two_rank.log
