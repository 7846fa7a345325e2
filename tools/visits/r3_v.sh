#!/bin/bash
# round 3, visit v: the co-resident small-cout convolution -- tests, kernel timing, in-step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3v
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -k "small_cout or conv2d_vs_torch or multi_head" 2>&1 | tail -4 > gpurun_out/r3v/tests.log
timeout 300 python - > gpurun_out/r3v/time.log 2>&1 <<'PY'
import torch
from eavsr_amd import ops
dev = torch.device("cuda:0")
def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (n, cin, cout, h, w) in [(2, 64, 6, 180, 320), (2, 18, 2, 180, 320), (2, 64, 6, 90, 160), (14, 64, 3, 720, 1280)]:
    x = torch.randn(n, cin, h, w, device=dev)
    wt, b = torch.randn(cout, cin, 3, 3, device=dev) * 0.05, torch.randn(cout, device=dev) * 0.1
    res = {}
    for lite in (False, True):
        ops.SMALLCO_LITE = lite; ops.SMALLCO_LITE_MIN_TILES = 0
        res[lite] = timed(lambda: ops.conv2d(x, wt, b), reps=10 if h > 500 else 50)
    print(f"{n}x{cin}x{h}x{w} -> {cout}: classic {res[False]:.1f} us, co-resident variant {res[True]:.1f} us")
PY
for v in classic lite classic lite; do
  EAVSR_SMALLCO=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('EAVSR_SMALLCO=$v', d['ms_per_step'], d['ms_per_step_median'], d['timed_output_max_abs_vs_eager'])" >> gpurun_out/r3v/ab.log
done
cat gpurun_out/r3v/tests.log gpurun_out/r3v/time.log gpurun_out/r3v/ab.log | grep -v amdgpu
