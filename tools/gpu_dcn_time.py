#!/usr/bin/env python3
"""Time the DCNv2 kernel variants at the bench's launch shapes (HIP events on the launch stream, median of REPS)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eavsr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
h, w = 180, 320
reps = int(os.environ.get("REPS", 20))
r = lambda *s: torch.randn(*s, device=dev)
w33, b = r(64, 64, 3, 3) * 0.05, r(64) * 0.1


def timed(fn):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for n in (2, 4):
    x = r(n, 64, h, w)
    for sigma in (0.5, 1.5, 4.0):
        off = r(n, 144, h, w) * sigma
        mask = torch.rand(n, 72, h, w, device=dev)
        heads = torch.cat([r(n, 32, h, w) * 0.25 + torch.tensor([1.0, 0, 0, 1.0], device=dev).repeat(8).view(1, 32, 1, 1),
                           r(n, 16, h, w) * sigma, r(n, 72, h, w)], 1)
        xil = ops.to_il8(x)
        px = n * h * w
        res = {}
        ops.set_dcn_mode("native")
        res["native"] = timed(lambda: ops.modulated_deform_conv2d(x, off, mask, w33, b, 1, 1, 1, 1, 8))
        ops.set_dcn_mode("bf16x9")
        res["x9(r1)"] = timed(lambda: ops.modulated_deform_conv2d(x, off, mask, w33, b, 1, 1, 1, 1, 8))
        ops.set_dcn_mode("native")
        for impl in ("il", "ws"):
            ops.set_dcn_il_impl(impl)
            for nprod in (6, 9):
                res[f"{impl}{nprod}"] = timed(lambda: ops.dcnv2_il(xil, off, mask, w33, b, 8, nprod=nprod))
                res[f"{impl}{nprod}_heads"] = timed(lambda: ops.dcnv2_il(xil, heads, None, w33, b, 8, nprod=nprod, heads=True))
        res["to_il8"] = timed(lambda: ops.to_il8(x))
        print(f"n={n} sigma={sigma}: " + "  ".join(f"{k} {v:.1f} us ({1376.0 * px / v / 1e3 / 8000:.3f} of HBM)" for k, v in res.items()), flush=True)
