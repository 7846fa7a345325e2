#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_hip_configs.py -m gpu -q -x 2>&1 | tail -5
