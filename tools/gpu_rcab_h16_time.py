#!/usr/bin/env python3
"""Timing of the one-launch RCAB convolutions (eavsr_rcab_convs_h16) against the two launches of the resident-weights kernel at
configs[2]'s (4 x 256 x 256) and configs[4]'s (1 x 540 x 960) sub-batch shapes, in rotation; bits compared first."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eavsr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
w1, w2 = torch.randn(64, 64, 3, 3, device=dev) / 24, torch.randn(64, 64, 3, 3, device=dev) / 24
b1, b2 = torch.randn(64, device=dev) * 0.1, torch.randn(64, device=dev) * 0.1


def med(fn, reps=9, inner=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    return sorted(ts)[len(ts) // 2]


for dt, (n, h, w) in (("bf16", (4, 256, 256)), ("fp16", (1, 540, 960)), ("bf16", (2, 180, 320))):
    x = ops.to_nhwc_h16(torch.randn(n, 64, h, w, device=dev), dt)

    def two():
        t = ops.conv3x3_c64_h16(x, w1, b1, relu=True)
        return ops.conv3x3_c64_h16(t, w2, b2, chan_partial=True)

    def one():
        return ops.rcab_convs_h16(x, w1, b1, w2, b2, chan_partial=True)

    same = torch.equal(two()[0], one()[0])
    res = {"two": [], "one": []}
    for _ in range(4):
        res["two"].append(med(two))
        res["one"].append(med(one))
    m = lambda v: sorted(v)[len(v) // 2]
    px = n * h * w
    fl = 2 * 2.0 * 64 * 64 * 9 * px
    print(f"{dt} {n}x64x{h}x{w}: bits {'same' if same else 'DIFFER'}; two launches {m(res['two']):.1f} us, one launch {m(res['one']):.1f} us "
          f"({fl / m(res['one']) / 1e6 / 1e6:.3f} PFLOP/s algorithmic = {fl / m(res['one']) / 1e6 / 2.5e9:.3f} of 2.5 PF; "
          f"HBM {2.0 * px * 128 / m(res['one']) / 1e6:.2f} TB/s)", flush=True)
