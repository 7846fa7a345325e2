#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6s
RES=1 timeout 600 python tools/gpu_x6s_diag.py 2>&1 | tee gpurun_out/r6s/res.txt | tail -8
timeout 600 python tools/gpu_x6s_diag.py 2>&1 | tee gpurun_out/r6s/plain.txt | tail -8
