#!/bin/bash
# round 3, visit p: two ranks on ONE GPU through the default (hybrid gloo + RCCL) process group: the control plane of bench.py --gpus N
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3p
timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-profile > gpurun_out/r3p/bench2.json 2> gpurun_out/r3p/bench2.err
echo "rc=$?"
tail -c 1500 gpurun_out/r3p/bench2.json
tail -5 gpurun_out/r3p/bench2.err
timeout 900 python -m pytest tests/test_hip_ops.py -q -k "dcnv2_vs_oracle" 2>&1 | tail -3
