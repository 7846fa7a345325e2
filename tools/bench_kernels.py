#!/usr/bin/env python3
"""Launch the hot kernels at the bench shapes a few times (for rocprofv3 --pmc passes)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eavsr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n, h, w = int(os.environ.get("N", 2)), 180, 320   # bench.py's default launches: two clips per sub-batch
reps = int(os.environ.get("REPS", 5))
which = os.environ.get("WHICH", "conv,convhr,dcn,warp").split(",")
r = lambda *s: torch.randn(*s, device=dev)
x64 = r(n, 64, h, w)
w33, b = r(64, 64, 3, 3) * 0.05, r(64) * 0.1
if "conv" in which:
    for mode in ("winograd4", "winograd", "direct"):
        ops.set_conv_mode(mode)
        for _ in range(reps):
            ops.conv2d(x64, w33, b, act="relu")
    ops.set_conv_mode("winograd4")
    w55, b120 = r(120, 64, 5, 5) * 0.02, r(120) * 0.1      # the predictor's 5x5 heads, F(2x2, 5x5)
    for _ in range(reps):
        ops.conv2d(x64, w55, b120)
if "convhr" in which:
    hr = r(n, 64, 4 * h, 4 * w)
    for _ in range(2):
        ops.conv2d(hr, w33, b, act="relu")
if "dcn" in which:
    ops.set_dcn_mode(os.environ.get("EAVSR_DCN_MODE", "native"))     # the NCHW kernel of round 1; the IL8 one is "dcnil" below
    off = r(n, 144, h, w) * float(os.environ.get("SIGMA", 1.5))
    mask = torch.rand(n, 72, h, w, device=dev)
    for _ in range(reps):
        ops.modulated_deform_conv2d(x64, off, mask, w33, b, 1, 1, 1, 1, 8)
    if "dcnil" in which:
        xil = ops.to_il8(x64)
        heads = torch.cat([r(n, 32, h, w) * 0.25 + torch.tensor([1.0, 0, 0, 1.0], device=dev).repeat(8).view(1, 32, 1, 1),
                           r(n, 16, h, w) * float(os.environ.get("SIGMA", 1.5)), r(n, 72, h, w)], 1)
        for impl in ("il", "il2"):       # round 2's schedule and round 4's: the counters of both in one pass
            ops.set_dcn_il_impl(impl)
            for nprod in (6, 9):
                for _ in range(reps):
                    ops.dcnv2_il(xil, off, mask, w33, b, 8, nprod=nprod)
                for _ in range(reps):
                    ops.dcnv2_il(xil, heads, None, w33, b, 8, nprod=nprod, heads=True)
if "warp" in which:
    flow = r(n, 2, h, w) * 2
    for _ in range(reps):
        ops.flow_warp(x64, flow)
    for _ in range(reps):
        ops.flow_warp_pair(x64, x64, flow, b_il8=True)
torch.cuda.synchronize()
print("done")
