#!/usr/bin/env python3
"""Per-kernel timing at the bench shapes (BASELINE.json configs[1]); prints as it goes."""
import faulthandler
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

faulthandler.dump_traceback_later(240, repeat=True, file=sys.stderr)
from eavsr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def T(name, fn, flops=0.0, nbytes=0.0, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    msg = f"{name:34s} {dt * 1e3:9.3f} ms"
    if flops:
        msg += f"  {flops / dt / 1e12:7.2f} TFLOP/s"
    if nbytes:
        msg += f"  {nbytes / dt / 1e9:8.1f} GB/s"
    print(msg, flush=True)
    return dt


n, h, w = int(os.environ.get("DIAG_N", 4)), 180, 320
px = n * h * w
r = lambda *s: torch.randn(*s, device=dev)
x64 = r(n, 64, h, w)
flow = r(n, 2, h, w) * 2
print("device", torch.cuda.get_device_name(0), "n", n, flush=True)
T("flow_warp c64", lambda: ops.flow_warp(x64, flow), nbytes=4 * px * 130)
w33 = r(64, 64, 3, 3) * 0.05
b = r(64) * 0.1
ops.set_conv_mode("direct")
T("conv3x3 64->64 relu", lambda: ops.conv2d(x64, w33, b, act="relu"), flops=2 * 64 * 64 * 9 * px)
T("conv3x3 64->64 +partial", lambda: ops.conv2d(x64, w33, b, chan_partial=True), flops=2 * 64 * 64 * 9 * px)
w5 = r(120, 64, 5, 5) * 0.02
T("conv5x5 64->120", lambda: ops.conv2d(x64, w5, None), flops=2 * 64 * 120 * 25 * px)
w1 = r(64, 192, 1, 1) * 0.05
T("conv1x1 192->64 (3 src)", lambda: ops.conv2d([x64, x64, x64], w1, b), flops=2 * 192 * 64 * px)
w320 = r(64, 320, 3, 3) * 0.02
T("conv3x3 320->64 (5 src)", lambda: ops.conv2d([x64] * 5, w320, b, act="lrelu", slope=0.1), flops=2 * 320 * 64 * 9 * px)
w18 = r(2, 18, 3, 3)
x18 = r(n, 18, h, w)
T("conv3x3 18->2", lambda: ops.conv2d(x18, w18, None), flops=2 * 18 * 2 * 9 * px)
w6 = r(6, 64, 3, 3)
T("conv3x3 64->6", lambda: ops.conv2d(x64, w6, None), flops=2 * 64 * 6 * 9 * px)
off = r(n, 144, h, w) * 1.5
mask = torch.rand(n, 72, h, w, device=dev)
T("dcnv2 sigma1.5", lambda: ops.modulated_deform_conv2d(x64, off, mask, w33, b, 1, 1, 1, 1, 8),
  flops=2 * 64 * 64 * 9 * px, nbytes=4 * px * 344)
off0 = off * 0
T("dcnv2 sigma0", lambda: ops.modulated_deform_conv2d(x64, off0, mask, w33, b, 1, 1, 1, 1, 8),
  flops=2 * 64 * 64 * 9 * px, nbytes=4 * px * 344)
ops.set_conv_mode("winograd")
T("conv3x3 64->64 relu  winograd", lambda: ops.conv2d(x64, w33, b, act="relu"), flops=2 * 64 * 64 * 9 * px)
T("conv3x3 64->64 +partial winograd", lambda: ops.conv2d(x64, w33, b, chan_partial=True), flops=2 * 64 * 64 * 9 * px)
ops.set_conv_mode("bf16x9")
T("conv3x3 64->64 relu  bf16x9", lambda: ops.conv2d(x64, w33, b, act="relu"), flops=2 * 64 * 64 * 9 * px)
T("conv3x3 64->64 +partial bf16x9", lambda: ops.conv2d(x64, w33, b, chan_partial=True), flops=2 * 64 * 64 * 9 * px)
ops.set_conv_mode(os.environ.get("EAVSR_CONV_MODE", "winograd"))
ops.set_dcn_mode("bf16x9")
T("dcnv2 bf16x9 sigma1.5", lambda: ops.modulated_deform_conv2d(x64, off, mask, w33, b, 1, 1, 1, 1, 8),
  flops=2 * 64 * 64 * 9 * px, nbytes=4 * px * 344)
T("dcnv2 bf16x9 sigma0", lambda: ops.modulated_deform_conv2d(x64, off0, mask, w33, b, 1, 1, 1, 1, 8),
  flops=2 * 64 * 64 * 9 * px, nbytes=4 * px * 344)
ops.set_dcn_mode(os.environ.get("EAVSR_DCN_MODE", "native"))
wa, ba, wb, bb = r(128, 1, 3, 3), r(128), r(64, 2, 3, 3), r(64)
T("adapt_frontend", lambda: ops.adapt_frontend(x64, x64, wa, ba, wb, bb), nbytes=4 * px * 192)
heads = r(n, 120, h, w)
T("affine_offsets D8", lambda: ops.affine_offsets(heads, 8, True), nbytes=4 * px * 336)
sc = torch.rand(n, 64, device=dev)
T("scale_residual", lambda: ops.scale_residual(x64, sc, x64), nbytes=4 * px * 192)
T("add3 (2ch)", lambda: ops.add(flow, flow, flow))
T("resize_ac 2ch /4", lambda: ops.resize_bilinear_ac(flow, (h // 4, w // 4), 0.25))
f28 = r(7 * n, 64, h, w)
T("pyramid 7n x 64", lambda: ops.pyramid(f28), nbytes=4 * 7 * px * 64 * 1.3125)
hr = r(n, 64, 4 * h, 4 * w)
T("conv3x3 64->64 @HR (n imgs)", lambda: ops.conv2d(hr, w33, b, act="lrelu", slope=0.1), flops=2 * 64 * 64 * 9 * px * 16, reps=2)
w3o = r(3, 64, 3, 3)
T("conv3x3 64->3 @HR", lambda: ops.conv2d(hr, w3o, None), flops=2 * 64 * 3 * 9 * px * 16, reps=2)
w256 = r(256, 64, 3, 3) * 0.05
T("conv3x3 64->256", lambda: ops.conv2d(x64, w256, None), flops=2 * 64 * 256 * 9 * px, reps=2)
x8 = r(2 * 6 * n, 8, 192, 320)
w7 = r(32, 8, 7, 7) * 0.05
T("conv7x7 8->32 (spynet L5)", lambda: ops.conv2d(x8, w7, None, act="relu"), flops=2 * 8 * 32 * 49 * 12 * n * 192 * 320, reps=2)
x32 = r(2 * 6 * n, 32, 192, 320)
w7b = r(64, 32, 7, 7) * 0.02
T("conv7x7 32->64 (spynet L5)", lambda: ops.conv2d(x32, w7b, None, act="relu"), flops=2 * 32 * 64 * 49 * 12 * n * 192 * 320, reps=2)

# ---- whole model, staged ----
from argparse import Namespace  # noqa: E402
from eavsr_amd.eavsrp_model import EAVSRP  # noqa: E402
from eavsr_amd.utils.synthetic import fill_state_dict, shapes_of, synthetic_clip  # noqa: E402
net = EAVSRP(Namespace(predict=False, n_frame=7, n_flow=5, scale=4), None)
sd0 = net.state_dict()
net.load_state_dict(fill_state_dict(shapes_of(sd0), "trained_like", fixed=sd0))
net = net.to(dev).eval()
clips = synthetic_clip(n, 7, h, w, 0).to(dev)
with torch.no_grad():
    t0 = time.perf_counter(); ff, fb = net.compute_flow(clips); torch.cuda.synchronize()
    print(f"compute_flow                       {1e3 * (time.perf_counter() - t0):9.1f} ms (first call)", flush=True)
    T("compute_flow", lambda: net.compute_flow(clips), reps=2)
    lr_tm = clips.transpose(0, 1).reshape(7 * n, 3, h, w)
    T("encoder", lambda: net.encoder(lr_tm), reps=2)
    for rep in range(2):
        t0 = time.perf_counter(); y = net(clips); torch.cuda.synchronize()
        print(f"full forward #{rep}                   {1e3 * (time.perf_counter() - t0):9.1f} ms -> {7 * n / (time.perf_counter() - t0):.2f} frames/s", flush=True)
    with ops.profile() as prof:
        net(clips)
    s = prof.summary()
    tot = sum(v["ms"] for v in s.values())
    print(f"instrumented step: {tot:.1f} ms of kernels", flush=True)
    for k, v in sorted(s.items(), key=lambda kv: -kv[1]["ms"])[:25]:
        print(f"  {k:26s} calls {v['calls']:5d}  {v['ms']:9.2f} ms  avg {v['ms'] / v['calls'] * 1e3:9.1f} us  "
              f"{v['flops'] / max(v['ms'], 1e-9) / 1e9:8.2f} TFLOP/s  {v['bytes'] / max(v['ms'], 1e-9) / 1e6:8.1f} GB/s", flush=True)
