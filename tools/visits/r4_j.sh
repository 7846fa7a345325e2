#!/bin/bash
# round 4, visit j: what-if timings of the step at HEAD (tools/gpu_whatif.py) + new GPU tests of this round
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4j
timeout 900 python3 tools/gpu_whatif.py > gpurun_out/r4j/whatif.log 2>&1
echo "exit $?" >> gpurun_out/r4j/whatif.log
cat gpurun_out/r4j/whatif.log | grep -v amdgpu.ids
timeout 1200 python3 -m pytest tests/test_hip_configs.py tests/test_hip_ops.py -q -x -m gpu -k "config1 or epilogue_relu or ca_out_without or pack_smallco" > gpurun_out/r4j/pytest_new.log 2>&1
tail -6 gpurun_out/r4j/pytest_new.log
