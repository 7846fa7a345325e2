#!/bin/bash
# round 3, visit w: small-tile shape of the co-resident small-cout convolution, in-step A/B by library
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3w
for v in default t464 t264 t832 default t464 t264 t832; do
  if [ $v = default ]; then unset EAVSR_LIB_PATH; else export EAVSR_LIB_PATH=$PWD/eavsr_amd/lib/libeavsr_$v.so; fi
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['ms_per_step_median'], d['timed_output_max_abs_vs_eager'])" >> gpurun_out/r3w/ab.log
done
cat gpurun_out/r3w/ab.log
