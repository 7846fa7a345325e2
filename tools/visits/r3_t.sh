#!/bin/bash
# round 3, visit t: DCNv2 operand split with packed residual subtractions -- tests and timing, A/B by library
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3t
EAVSR_LIB_PATH=$PWD/eavsr_amd/lib/libeavsr_pk.so timeout 900 python -m pytest tests/test_hip_ops.py -x -q -k "dcnv2_il" 2>&1 | tail -3 > gpurun_out/r3t/tests.log
for v in default pk default pk; do
  if [ $v = default ]; then unset EAVSR_LIB_PATH; else export EAVSR_LIB_PATH=$PWD/eavsr_amd/lib/libeavsr_$v.so; fi
  echo "== $v" >> gpurun_out/r3t/time.log
  REPS=40 timeout 600 python tools/gpu_dcn_time.py 2>/dev/null | grep "n=2" | sed 's/  /\n   /g' | grep -E "^n=|il6|il9" >> gpurun_out/r3t/time.log
done
cat gpurun_out/r3t/tests.log gpurun_out/r3t/time.log
