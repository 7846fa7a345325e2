#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6r
ONLY=aten::zero_,aten::fill_,aten::zeros,aten::zeros_like TOP=400 timeout 600 python tools/dbg/train_glue.py > gpurun_out/r6r/glue.txt 2>&1
tail -70 gpurun_out/r6r/glue.txt
