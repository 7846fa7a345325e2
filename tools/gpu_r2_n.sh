#!/bin/bash
# round 2, visit n: the 16-bit 5x5 heads kernel -- op tests, 16-bit model tests, timing, bench in bf16 / fp16 mode
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2n
timeout 900 python -m pytest tests/test_hip_h16.py -q -x 2>&1 | tail -15 > gpurun_out/r2n/h16.log
timeout 900 python -m pytest tests/test_hip_configs.py -q -s -k "config4_fifteen or config2" 2>&1 | grep -v amdgpu | tail -8 > gpurun_out/r2n/configs.log
timeout 300 python - > gpurun_out/r2n/time.log 2>&1 <<'PY'
import torch
from eavsr_amd import ops
dev = torch.device("cuda:0")
n, h, w = 2, 180, 320
f = torch.randn(n, 64, h, w, device=dev)
ws = [torch.randn(c, 64, 5, 5, device=dev) * 0.02 for c in (32, 16, 72)]
bs = [torch.randn(c, device=dev) * 0.1 for c in (32, 16, 72)]
def timed(fn, reps=20):
    for _ in range(3): fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
for dt in ("bf16", "fp16"):
    fh = ops.to_nhwc_h16(f, dt)
    t_conv = timed(lambda: ops.conv5x5_c64_h16(fh, ws, bs))
    t_cvt = timed(lambda: ops.to_nhwc_h16(f, dt))
    fl = 2.0 * 64 * 120 * 25 * n * h * w
    print(f"{dt}: conv5x5_c64_h16 {t_conv:.1f} us ({fl / t_conv / 1e6:.0f} TFLOP/s), to_nhwc_h16 {t_cvt:.1f} us")
t32 = timed(lambda: ops.conv2d(f, ws, bs, padding=2) if hasattr(ops, 'conv2d') else None)
print(f"fp32 heads (Winograd F(2x2,5x5)): {t32:.1f} us")
PY
for dt in bf16 fp16; do timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --backbone-dtype $dt > gpurun_out/r2n/bench_$dt.json 2> gpurun_out/r2n/bench_$dt.err; done
cat gpurun_out/r2n/h16.log gpurun_out/r2n/configs.log gpurun_out/r2n/time.log
python - <<'PY'
import json
for f in ("bf16", "fp16"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r2n/bench_{f}.json") if l.startswith("{")][-1])
        print(f, d["ms_per_step"], d["value"], d.get("step_breakdown_ms"), d.get("share_of_step_in_16bit"))
    except Exception as e:
        print(f, "failed", e)
PY
