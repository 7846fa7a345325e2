#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6x
bash tools/gpu_train_prof.sh > gpurun_out/r6x/train_prof.log 2>&1
head -60 gpurun_out/r6x/train_prof.log | cut -c1-170
