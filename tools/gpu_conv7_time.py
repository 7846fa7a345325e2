#!/usr/bin/env python3
"""SPyNet 7x7 layers (eavsrp_model.py:398-431) at the pyramid shapes of the bench, and the 5x5 heads: eavsr_conv_f32x6 against the fp32-MFMA kernels."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eavsr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
N = int(os.environ.get("N", 24))     # image pairs of a 2-clip sub-batch (2 x 6 pairs x 2 directions)
tot = {"bf16x6": 0.0, "fp32": 0.0}
for lvl, (h, w) in enumerate([(6, 10), (12, 20), (24, 40), (48, 80), (96, 160), (192, 320)]):
    for cin, cout in [(8, 32), (32, 64), (64, 32), (32, 16), (16, 2)]:
        x = torch.randn(N, cin, h, w, device=dev)
        wt = torch.randn(cout, cin, 7, 7, device=dev) * 0.02
        b = torch.randn(cout, device=dev) * 0.1
        res = {}
        for mode in ("bf16x6", "fp32"):
            ops.CONV7_MODE = mode
            for _ in range(2):
                y = ops.conv2d(x, wt, b, act="relu")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 5
            e0.record()
            for _ in range(reps):
                y = ops.conv2d(x, wt, b, act="relu")
            e1.record()
            torch.cuda.synchronize()
            res[mode] = e0.elapsed_time(e1) / reps * 1e3
            tot[mode] += res[mode]
            res[mode + "_y"] = y
        d = (res["bf16x6_y"] - res["fp32_y"]).abs().max().item()
        fl = 2.0 * cin * cout * 49 * N * h * w
        print(f"level {lvl} {h:3d}x{w:3d} {cin:2d}->{cout:2d}: bf16x6 {res['bf16x6']:8.1f} us ({fl / res['bf16x6'] / 1e6:6.1f} TFLOP/s)   "
              f"fp32 {res['fp32']:8.1f} us ({fl / res['fp32'] / 1e6:6.1f} TFLOP/s)   max |diff| {d:.2e}", flush=True)
print(f"sum over the pyramid: bf16x6 {tot['bf16x6'] / 1e3:.2f} ms   fp32 {tot['fp32'] / 1e3:.2f} ms")
# the predictor's 5x5 heads (networks.py:289-315): one 64 -> 120 launch per frame step of a 2-clip sub-batch
x = torch.randn(2, 64, 180, 320, device=dev)
ws = [torch.randn(co, 64, 5, 5, device=dev) * 0.02 for co in (32, 16, 72)]
bs = [torch.randn(co, device=dev) * 0.1 for co in (32, 16, 72)]
res = {}
for mode in ("bf16x6", "wino"):
    ops.CONV5_MODE = mode
    for _ in range(3):
        y = ops.conv2d(x, ws, bs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = ops.conv2d(x, ws, bs)
    e1.record()
    torch.cuda.synchronize()
    res[mode] = (e0.elapsed_time(e1) / 20 * 1e3, y)
print(f"5x5 heads 2 x 64 x 180 x 320 -> 120: bf16x6 {res['bf16x6'][0]:.1f} us   F(2x2,5x5) fp32 {res['wino'][0]:.1f} us   "
      f"max |diff| {(res['bf16x6'][1] - res['wino'][1]).abs().max().item():.2e}")
