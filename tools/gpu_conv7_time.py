#!/usr/bin/env python3
"""SPyNet's 7x7 layers (eavsrp_model.py:398-431) at the bench's pyramid shapes: eavsr_conv7x7_f32x6 against the fp32-MFMA kernel."""
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
