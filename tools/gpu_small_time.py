#!/usr/bin/env python3
"""Time the small streaming kernels of the predictor / tail at the bench's launch shapes (back to back, HIP events)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eavsr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
r = lambda *s: torch.randn(*s, device=dev)


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


n, h, w = 2, 180, 320
x = r(n, 64, h, w)
hh = r(n, 64, h, w)
res = {}
for cout, cin, hh_, ww_, nn in ((6, 64, h, w, n), (6, 64, h // 2, w // 2, n), (2, 18, h, w, n), (3, 64, 720, 1280, 2)):
    xi = r(nn, cin, hh_, ww_)
    wt = r(cout, cin, 3, 3) * 0.05
    b = r(cout)
    res[f"conv3x3 {cin}->{cout} {nn}x{hh_}x{ww_}"] = timed(lambda: ops.conv2d(xi, wt, b), reps=10 if hh_ > 500 else 30)
w1, b1, w2, b2 = r(128, 1, 3, 3) * 0.2, r(128), r(64, 2, 3, 3) * 0.2, r(64)
res["adapt_frontend 2x64x180x320"] = timed(lambda: ops.adapt_frontend(x, hh, w1, b1, w2, b2))
sc = torch.rand(n, 64, device=dev)
res["scale_residual 2x64x180x320"] = timed(lambda: ops.scale_residual(x, sc, hh))
tiles = ((h + 7) // 8) * ((w + 63) // 64)
part = r(n, tiles, 64)
cw1, cb1, cw2, cb2 = r(4, 64, 1, 1), r(4), r(64, 4, 1, 1), r(64)
res["ca_scale 2 x 115 tiles x 64"] = timed(lambda: ops.ca_scale(part, h * w, cw1, cb1, cw2, cb2))
fl = r(n, 2, h, w)
res["flow_warp 2x64x180x320"] = timed(lambda: ops.flow_warp(x, fl))
for k, v in res.items():
    print(f"{k:40s} {v:8.1f} us", flush=True)
