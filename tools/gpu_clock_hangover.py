#!/usr/bin/env python3
"""Does a bf16-MFMA-heavy launch (the 5x5 heads by eavsr_conv_f32x6, which runs the chip into its power limit: DESIGN 4l) slow the
fp32 Winograd convolutions that follow it on the same stream?  Times K 3x3 launches by events, alone and right behind one heads launch
in either mode."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eavsr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
x = torch.randn(2, 64, 180, 320, device=dev)
w3 = torch.randn(64, 64, 3, 3, device=dev) * 0.05
b3 = torch.randn(64, device=dev) * 0.1
ws = [torch.randn(co, 64, 5, 5, device=dev) * 0.02 for co in (32, 16, 72)]
bs = [torch.randn(co, device=dev) * 0.1 for co in (32, 16, 72)]


def run(mode5, k, reps=30):
    ops.CONV5_MODE = mode5 if mode5 else "bf16x6"
    tot, tot5 = 0.0, 0.0
    for _ in range(3):
        ops.conv2d(x, ws, bs); ops.conv2d(x, w3, b3, act="relu")
    torch.cuda.synchronize()
    for _ in range(reps):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        if mode5:
            ops.conv2d(x, ws, bs)
        e[1].record()
        for _ in range(k):
            ops.conv2d(x, w3, b3, act="relu")
        e[2].record()
        torch.cuda.synchronize()
        tot5 += e[0].elapsed_time(e[1]); tot += e[1].elapsed_time(e[2])
    return tot5 / reps * 1e3, tot / reps / k * 1e3


for k in (2, 5, 10, 40):
    a = run(None, k)
    b = run("bf16x6", k)
    c = run("wino", k)
    print(f"{k:3d} 3x3 launches: alone {a[1]:6.2f} us each | behind a bf16x6 heads launch ({b[0]:6.1f} us) {b[1]:6.2f} us each | "
          f"behind an F(2x2,5x5) heads launch ({c[0]:6.1f} us) {c[1]:6.2f} us each", flush=True)
