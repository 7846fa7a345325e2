#!/bin/bash
# round 3, visit x: full GPU suite at the new defaults
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3x
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r3x/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3x/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r3x/smoke.log
cat gpurun_out/r3x/tests.log; tail -4 gpurun_out/r3x/smoke.log
