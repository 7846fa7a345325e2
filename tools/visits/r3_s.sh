#!/bin/bash
# round 3, visit s: the RCAB tail as ONE launch of small workgroups -- bit-identity test, kernel timing, in-step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3s
timeout 600 python -m pytest tests/test_hip_ops.py -x -q -k "ca_tail or channel_attention" 2>&1 | tail -3 > gpurun_out/r3s/tests.log
timeout 300 python - > gpurun_out/r3s/time.log 2>&1 <<'PY'
import torch
from eavsr_amd import ops
dev = torch.device("cuda:0")
n, c, h, w, tiles = 2, 64, 180, 320, 115
r, x = torch.randn(n, c, h, w, device=dev), torch.randn(n, c, h, w, device=dev)
part = torch.randn(n, tiles, c, device=dev)
w1, b1, w2, b2 = torch.randn(4, 64, 1, 1, device=dev) * 0.2, torch.randn(4, device=dev) * 0.1, torch.randn(64, 4, 1, 1, device=dev) * 0.5, torch.randn(64, device=dev) * 0.1
def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
t_two = timed(lambda: ops.scale_residual(r, ops.ca_scale(part, h * w, w1, b1, w2, b2), x))
t_one = timed(lambda: ops.ca_tail(r, part, w1, b1, w2, b2, x))
t_sr = timed(lambda: ops.scale_residual(r, torch.ones(n, c, device=dev), x))
print(f"ca_scale + scale_residual {t_two:.1f} us; scale_residual alone {t_sr:.1f} us; ca_tail (one launch) {t_one:.1f} us   [back to back, incl. host launch gaps]")
PY
for v in 0 1 0 1; do
  EAVSR_FUSE_CA_TAIL=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('EAVSR_FUSE_CA_TAIL=$v', d['ms_per_step'], d['ms_per_step_median'], d['timed_output_max_abs_vs_eager'])" >> gpurun_out/r3s/ab.log
done
cat gpurun_out/r3s/tests.log gpurun_out/r3s/time.log gpurun_out/r3s/ab.log | grep -v amdgpu
