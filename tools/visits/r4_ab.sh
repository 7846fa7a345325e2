#!/bin/bash
# round 4: the one-launch RCAB tail (EAVSR_FUSE_CA_TAIL=1) against the two-launch default at HEAD, A/B twice; stream count 2 / 3
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4ab
run() { timeout 300 python3 bench.py --no-cpu-baseline --no-kernel-profile --also '' --steps 10 "$@" | python3 -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['ms_per_step'], l['timed_output_check']['bit_identical'])"; }
{
for r in 1 2; do
  echo -n "default    "; run
  echo -n "fused tail "; EAVSR_FUSE_CA_TAIL=1 run
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r4ab/log.txt
cat gpurun_out/r4ab/log.txt
