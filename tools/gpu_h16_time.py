#!/usr/bin/env python3
"""Time the 16-bit backbone kernels at the bench's launch shape (2 x 180 x 320 x 64 NHWC), back to back."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eavsr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n, h, w = int(os.environ.get("N", 2)), 180, 320
x = torch.randn(n, 64, h, w, device=dev)
wt = torch.randn(64, 64, 3, 3, device=dev) * 0.04
b = torch.randn(64, device=dev) * 0.1


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for dt in ("bf16", "fp16"):
    xh = ops.to_nhwc_h16(x, dt)
    sc = torch.rand(n, 64, device=dev)
    res = {
        "conv": timed(lambda: ops.conv3x3_c64_h16(xh, wt, b)),
        "conv+relu": timed(lambda: ops.conv3x3_c64_h16(xh, wt, b, relu=True)),
        "conv+chan_partial": timed(lambda: ops.conv3x3_c64_h16(xh, wt, b, chan_partial=True)),
        "scale_residual_h16": timed(lambda: ops.scale_residual_h16(xh, sc, xh)),
        "to_nhwc_h16": timed(lambda: ops.to_nhwc_h16(x, dt)),
    }
    fl = 2.0 * 64 * 64 * 9 * n * h * w
    print(f"{dt} n={n}: " + "  ".join(f"{k} {v:.1f} us" for k, v in res.items()) + f"   (conv: {fl / res['conv'] / 1e6:.0f} TFLOP/s, "
          f"{2.0 * n * h * w * 128 / res['conv'] / 1e3:.0f} GB/s)", flush=True)
