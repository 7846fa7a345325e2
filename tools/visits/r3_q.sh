#!/bin/bash
# round 3, visit q: L2 warm-up of the launch's weights -- microbenchmark (hot and rotating weights / inputs) and in-step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3q
NS=2 timeout 300 python tools/gpu_wino4_timeline.py > gpurun_out/r3q/timeline.log 2>&1
for v in default pf default pf; do
  if [ $v = default ]; then unset EAVSR_LIB_PATH; else export EAVSR_LIB_PATH=$PWD/eavsr_amd/lib/libeavsr_$v.so; fi
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['ms_per_step_median'], d['timed_output_max_abs_vs_eager'])" >> gpurun_out/r3q/ab.log
done
grep -v amdgpu.ids gpurun_out/r3q/timeline.log | grep -v "^    \|^  \["
cat gpurun_out/r3q/ab.log
