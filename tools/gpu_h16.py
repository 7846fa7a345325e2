#!/usr/bin/env python3
"""Timing of the 16-bit backbone kernels at the bench shape and of the whole forward in 16-bit backbone mode."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from eavsr_amd import ops, networks as Nw
dev = torch.device("cuda:0")
def T(name, fn, flops=0.0, nbytes=0.0, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:40s} {dt*1e6:9.1f} us  {flops/dt/1e12:8.1f} TFLOP/s  {nbytes/dt/1e9:8.1f} GB/s", flush=True)
for (n, h, w) in [(4, 180, 320), (8, 256, 256), (1, 540, 960)]:
    px = n * h * w
    for dt in (torch.bfloat16, torch.float16):
        x = torch.randn(n, h, w, 64, device=dev).to(dt)
        wt = torch.randn(64, 64, 3, 3, device=dev) * 0.04
        b = torch.randn(64, device=dev) * 0.1
        T(f"conv3x3_c64_h16 {dt} {n}x{h}x{w}", lambda: ops.conv3x3_c64_h16(x, wt, b, relu=True), 2 * 64 * 64 * 9 * px, 256.0 * px)
        T(f"  + chan_partial", lambda: ops.conv3x3_c64_h16(x, wt, b, chan_partial=True), 2 * 64 * 64 * 9 * px, 256.0 * px)
    sc = torch.rand(n, 64, device=dev)
    T(f"scale_residual_h16 {n}x{h}x{w}", lambda: ops.scale_residual_h16(x, sc, x), 0, 384.0 * px)
    xf = torch.randn(n, 64, h, w, device=dev)
    T(f"to_nhwc_h16", lambda: ops.to_nhwc_h16(xf, "bf16"), 0, 384.0 * px)
    T(f"from_nhwc_h16", lambda: ops.from_nhwc_h16(x, residual=xf), 0, 640.0 * px)
from argparse import Namespace
from eavsr_amd.eavsrp_model import EAVSRP
from eavsr_amd.utils.synthetic import fill_state_dict, shapes_of, synthetic_clip
net = EAVSRP(Namespace(predict=False, n_frame=7, n_flow=5, scale=4), None)
sd0 = net.state_dict()
net.load_state_dict(fill_state_dict(shapes_of(sd0), "trained_like", fixed=sd0))
net = net.to(dev).eval()
clips = synthetic_clip(4, 7, 180, 320, 0).to(dev)
with torch.no_grad():
    ref = net(clips)
    for mode in ("bf16", "fp16"):
        Nw.set_backbone_dtype(mode)
        net(clips); torch.cuda.synchronize()
        t0 = time.perf_counter(); y = net(clips); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        err = (y - ref).abs().max().item()
        a = (y.clamp(0, 1) * 255).round(); b_ = (ref.clamp(0, 1) * 255).round()
        mse = ((a - b_) / 255).pow(2).mean().item()
        import math
        print(f"forward, {mode} backbone: {dt*1e3:.1f} ms -> {28/dt:.1f} frames/s; max|y - fp32| {err:.2e}; PSNR vs fp32 {(-10*math.log10(mse)) if mse > 0 else float('inf'):.1f} dB", flush=True)
        with ops.profile() as prof:
            net(clips)
        s = prof.summary()
        for k, v in sorted(s.items(), key=lambda kv: -kv[1]["ms"])[:8]:
            print(f"    {k:26s} calls {v['calls']:5d}  {v['ms']:8.2f} ms  avg {v['ms']/v['calls']*1e3:8.1f} us", flush=True)
    Nw.set_backbone_dtype(None)
