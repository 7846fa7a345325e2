#!/bin/bash
# round 3, visit ax: the training step (configs[3] per-GPU share: 2 clips x 7 x 3 x 96 x 96) at HEAD, graphed and eager
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3ax
for g in "--graph" ""; do
  for m in "bf16x6 bf16x6" "fp32 wino"; do
    set -- $m
    EAVSR_CONV7=$1 EAVSR_CONV5=$2 timeout 900 python bench.py --mode train $g --steps 8 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$g', '$m', d['ms_per_step'], d['value'], d['config'].get('launch'), d.get('loss'))" >> gpurun_out/r3ax/train.log
  done
done
cat gpurun_out/r3ax/train.log
