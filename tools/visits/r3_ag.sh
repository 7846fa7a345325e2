cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3ae
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -k "conv5x5" 2>&1 | grep -v "^$" | grep "^E \|passed\|failed" | head -40
