#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3j
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r3j/tests.log
cat gpurun_out/r3j/tests.log
