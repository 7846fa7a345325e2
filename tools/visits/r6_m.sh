#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6m
timeout 300 python tools/dbg/dcn_bwd_check.py 2>&1 | grep "sampler - columns" | head -8
timeout 300 python tools/dbg/dcn_bwd_time.py 2>&1 | tee gpurun_out/r6m/ablate.txt
