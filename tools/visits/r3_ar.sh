#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_hip_ops.py -m gpu -q -k "bf16x6" 2>&1 | tail -8
