#!/bin/bash
# round 3, visit aw: ca_scale with sixteen loads in flight (configs[4]: 1020 tiles per sample): parity + configs[4] step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3aw
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -k "ca_tail or channel_attention" 2>&1 | tail -3
timeout 900 python bench.py --config 4 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config4', d['ms_per_step'], d['value'], d['step_breakdown_ms'].get('ca_scale'))"
