#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6r
TOP=70 timeout 600 python tools/dbg/train_glue.py > gpurun_out/r6r/glue.txt 2>&1
tail -70 gpurun_out/r6r/glue.txt
