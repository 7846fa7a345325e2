#!/bin/bash
# round 5, visit f: the whole GPU suite at HEAD, then smoke()
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r5_f_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5_f_tests.log
tail -5 gpurun_out/r5_f_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_f_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r5_f_smoke.log
