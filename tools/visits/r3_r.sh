#!/bin/bash
# round 3, visit r: cheaper tile set-up in conv_wino6 -- op tests, in-step A/B against the previous build
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3r
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -k "conv or wino" 2>&1 | tail -3 > gpurun_out/r3r/tests.log
for v in default prev default prev; do
  if [ $v = default ]; then unset EAVSR_LIB_PATH; else export EAVSR_LIB_PATH=$PWD/eavsr_amd/lib/libeavsr_$v.so; fi
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['ms_per_step_median'], d['timed_output_max_abs_vs_eager'])" >> gpurun_out/r3r/ab.log
done
cat gpurun_out/r3r/tests.log gpurun_out/r3r/ab.log
