#!/usr/bin/env python3
"""Run the other BASELINE.json configs once (shape / memory robustness and timing):
configs[2] x2 model 8 x 7 x 3 x 256 x 256 (fp32 and bf16 backbone), configs[4] x4 1 x 15 x 3 x 540 x 960 (fp32 and fp16 backbone);
fp32 = the default mode (Winograd 3x3 convolutions), then the direct-convolution and bf16x9 modes."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from argparse import Namespace
from eavsr_amd import networks as Nw
from eavsr_amd.eavsrp_model import EAVSRP
from eavsr_amd.utils.synthetic import fill_state_dict, shapes_of, synthetic_clip
dev = torch.device("cuda:0")
def model(scale):
    net = EAVSRP(Namespace(predict=False, n_frame=7, n_flow=5, scale=scale), None)
    sd0 = net.state_dict()
    net.load_state_dict(fill_state_dict(shapes_of(sd0), "trained_like", fixed=sd0))
    return net.to(dev).eval()
def psnr(a, b):
    mse = (((a.clamp(0, 1) * 255).round() - (b.clamp(0, 1) * 255).round()) / 255).pow(2).mean().item()
    return float("inf") if mse == 0 else -10 * math.log10(mse)
for name, scale, (n, t, h, w), mode in [("configs[2] x2 8x7x256x256", 2, (8, 7, 256, 256), "bf16"),
                                         ("configs[4] x4 1x15x540x960", 4, (1, 15, 540, 960), "fp16")]:
    net = model(scale)
    clips = synthetic_clip(n, t, h, w, 0).to(dev)
    res = {}
    for m in (None, mode):
        Nw.set_backbone_dtype(m)
        with torch.no_grad():
            torch.cuda.reset_peak_memory_stats()
            y = net(clips); torch.cuda.synchronize()
            t0 = time.perf_counter(); y = net(clips); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res[m] = y
        print(f"{name} backbone={m or 'fp32'}: {dt*1e3:.0f} ms -> {n*t/dt:.1f} frames/s, out {tuple(y.shape)}, "
              f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB, finite={bool(torch.isfinite(y).all())}", flush=True)
    print(f"   {mode} vs fp32: max abs {float((res[mode]-res[None]).abs().max()):.2e}, PSNR {psnr(res[mode], res[None]):.1f} dB", flush=True)
    # the other fp32 modes: direct convolutions, and the exact-split (bf16x9) contractions
    Nw.set_backbone_dtype(None)
    from eavsr_amd import ops
    for cm, dm in (("winograd", "native"), ("direct", "native"), ("bf16x9", "bf16x9")):
        ops.set_conv_mode(cm); ops.set_dcn_mode(dm)
        with torch.no_grad():
            y = net(clips); torch.cuda.synchronize()
            t0 = time.perf_counter(); y = net(clips); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{name} fp32, conv {cm} / dcn {dm}: {dt*1e3:.0f} ms -> {n*t/dt:.1f} frames/s, max abs vs default {float((y-res[None]).abs().max()):.2e}", flush=True)
    ops.set_conv_mode("winograd4"); ops.set_dcn_mode("il6")
    Nw.set_backbone_dtype(None)
    del net, clips, res, y
    torch.cuda.empty_cache()
