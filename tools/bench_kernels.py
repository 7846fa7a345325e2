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
        try:
            ops.set_conv_mode(mode)
            for _ in range(reps):
                ops.conv2d(x64, w33, b, act="relu")
        except ops.LabBuildRequired:      # F(2x2,3x3) exists in the lab library only (python -m eavsr_amd.build --lab)
            pass
    ops.set_conv_mode("winograd4")
    w55, b120 = r(120, 64, 5, 5) * 0.02, r(120) * 0.1      # the predictor's 5x5 heads, F(2x2, 5x5)
    for _ in range(reps):
        ops.conv2d(x64, w55, b120)
if "rcab" in which:
    # one RCAB of the default fp32 path (networks.py RCABlock): conv + ReLU with channel sums and border pieces, the ONE small launch
    # (attention before the second convolution), the second convolution with x + scale * r as its epilogue
    wa, ba, wb, bb = r(4, 64, 1, 1) * 0.1, r(4) * 0.1, r(64, 4, 1, 1) * 0.1, r(64) * 0.1
    w33b = r(64, 64, 3, 3) * 0.05
    for _ in range(reps):
        t, part, pieces = ops.conv2d(x64, w33, b, act="relu", chan_partial=True, border=True)
        scale = ops.ca_scale_pre(t, part, w33b, b, wa, ba, wb, bb, border=pieces)
        ops.conv2d(t, w33b, b, residual=x64, res_scale=scale)
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
                if impl == "il2" and nprod == 6:      # what the bench step launches: masks activated by the heads' epilogue (heads = 2)
                    heads_act = torch.cat([heads[:, :48], torch.sigmoid(heads[:, 48:])], 1)
                    for _ in range(reps):
                        ops.dcnv2_il(xil, heads_act, None, w33, b, 8, nprod=nprod, heads=True, mask_activated=True)
if "warp" in which:
    flow = r(n, 2, h, w) * 2
    for _ in range(reps):
        ops.flow_warp(x64, flow)
    for _ in range(reps):
        ops.flow_warp_pair(x64, x64, flow, b_il8=True)
if "h16" in which:
    # the 16-bit modes' kernels at configs[2]'s sub-batch shape (4 x 64 x 256 x 256 bf16): backbone conv, generic 3x3 (320 -> 64 over
    # five sources, 128 -> 256), predictor heads, one conv + PixelShuffle(2) stage, conv_last
    hn, hh, hw = 4, 256, 256
    xs = [r(hn, 64, hh, hw) for _ in range(5)]
    x16 = ops.to_nhwc_h16(xs[0], "bf16")
    w320, w256 = r(64, 320, 3, 3) * 0.02, r(256, 128, 3, 3) * 0.03
    x128 = r(hn, 128, hh, hw)
    wps, bps = r(256, 64, 3, 3) * 0.05, r(256) * 0.1
    wl, bl = r(3, 64, 3, 3) * 0.05, r(3) * 0.1
    heads_w = [r(32, 64, 5, 5) * 0.02, r(16, 64, 5, 5) * 0.02, r(72, 64, 5, 5) * 0.02]
    heads_b = [r(32) * 0.1, r(16) * 0.1, r(72) * 0.1]
    ops.set_conv3_h16("bf16")
    with torch.no_grad():
        for _ in range(reps):
            ops.conv3x3_c64_h16(x16, w33, b, relu=True)
            ops.conv2d(xs, w320, b, act="lrelu", slope=0.1)
            ops.conv2d(x128, w256, None)
            ops.conv5x5_c64_h16(x16, heads_w, heads_b)
            y16 = ops.conv3x3_c64_h16_act(x16, wps, bps, act="lrelu", slope=0.1, pixel_shuffle2=True)
            ops.conv3x3_c64to3_h16(y16, wl, bl)
    ops.set_conv3_h16(None)
if "h16b" in which:
    # ONE launch shape per kernel name (the h16 set averages the backbone launch with the four-slice pixel-shuffle
    # launch): the 16-bit backbone convolution at configs[2]'s sub-batch shape, and the one-launch RCAB convolutions (csrc/rcab_h16.hip)
    hn, hh, hw = 4, 256, 256
    x16 = ops.to_nhwc_h16(r(hn, 64, hh, hw), "bf16")
    w33b = r(64, 64, 3, 3) * 0.05
    with torch.no_grad():
        for _ in range(reps):
            t16 = ops.conv3x3_c64_h16(x16, w33, b, relu=True)
            ops.conv3x3_c64_h16(t16, w33b, b, chan_partial=True)
        try:
            for _ in range(reps):
                ops.rcab_convs_h16(x16, w33, b, w33b, b, chan_partial=True)
        except ops.LabBuildRequired:      # the one-launch RCAB convolutions: lab library only
            pass
if "train" in which:
    # the training step's two dominant kernels at configs[3]'s launch shapes: the crop-sized 3x3 64 -> 64 convolution (2 x 64 x 96 x 96)
    # and the 3x3 weight gradient over the seven frames of a clip as segments of one launch
    xt = r(2, 64, 96, 96)
    for _ in range(reps):
        ops.conv2d(xt, w33, b, act="relu")
    dys = [r(2, 64, 96, 96) for _ in range(7)]
    xs = [[r(2, 64, 96, 96)] for _ in range(7)]
    dw = torch.empty(64, 64, 3, 3, device=dev)
    for _ in range(reps):
        ops.conv_wgrad_multi(dys, xs, 3, out=dw)
torch.cuda.synchronize()
print("done")
