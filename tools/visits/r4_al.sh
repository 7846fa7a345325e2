#!/bin/bash
# round 4: grouped 3x3 convolution kernels of the training path with their nine taps loaded in one batch: backward tests, training step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4al
{
timeout 1500 python3 -m pytest tests/test_hip_backward.py -x -q -m gpu 2>&1 | tail -2
for r in 1 2; do
timeout 600 python3 bench.py --mode train --steps 10 --warmup 3 | python3 -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train', round(l['ms_per_step'],2), l['loss'])"
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r4al/log.txt
cat gpurun_out/r4al/log.txt
