#!/bin/bash
# round 6, evidence visit: profile of the bench (kernel trace + PMC passes + traffic + the default bench line), the training
# step's census and line, the 16-bit kernels' PMC blocks
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
COMMIT=$1 bash tools/gpu_profile.sh r06 > gpurun_out/r6_n_profile.log 2>&1
tail -3 gpurun_out/r6_n_profile.log
bash tools/gpu_train_prof.sh > gpurun_out/r6_n_train_prof.log 2>&1
head -3 gpurun_out/r6_n_train_prof.log
timeout 600 python bench.py --mode train --steps 5 --warmup 2 > gpurun_out/r6_n_train.json 2> gpurun_out/r6_n_train.err
tail -c 600 gpurun_out/r6_n_train.json
bash tools/gpu_profile_h16.sh > gpurun_out/r6_n_h16.log 2>&1
tail -25 gpurun_out/r6_n_h16.log
