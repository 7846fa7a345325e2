#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6o
timeout 300 python tools/dbg/dcn_bwd_check.py 2>&1 | grep "sampler - columns" | head -8
timeout 300 python tools/dbg/dcn_bwd_time.py 2>&1 | tee gpurun_out/r6o/ablate.txt
timeout 900 python -m pytest tests/test_hip_backward.py -m gpu -x -q -k "dcn" 2>&1 | tail -3
