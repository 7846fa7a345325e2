#!/usr/bin/env python3
"""eavsr_dcnv2_bwd_f32 (the sampler-side backward) against the column path (im2col -> GEMMs -> col2im) at the training shape, and timing"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from eavsr_amd import ops, autograd as AG
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (n, h, w, sigma) in ((2, 96, 96, 1.0), (2, 96, 96, 6.0), (1, 45, 77, 2.0), (3, 7, 5, 1.0)):
    x = torch.randn(n, 64, h, w, device=dev); off = torch.randn(n, 144, h, w, device=dev) * sigma
    mask = torch.rand(n, 72, h, w, device=dev); wt = torch.randn(64, 64, 3, 3, device=dev) / 24; dy = torch.randn(n, 64, h, w, device=dev)
    res = {}
    for mode in ("columns", "sampler"):
        ops.DCN_BWD = mode
        xs, os_, ms, ws = (t.clone().requires_grad_(True) for t in (x, off, mask, wt))
        out = AG.modulated_deform_conv2d(xs, os_, ms, ws, None, 1, 1, 1, 1, 8)
        res[mode] = torch.autograd.grad((out * dy).sum(), [xs, os_, ms, ws])
    for nme, a, b in zip(("dx", "doffset", "dmask", "dweight"), res["columns"], res["sampler"]):
        sc = max(1e-6, a.abs().max().item())
        print(f"{n}x{h}x{w} sigma {sigma}: {nme:8s} max|sampler - columns| / scale = {(a - b).abs().max().item() / sc:.2e}")
    if (n, h, w) == (2, 96, 96):
        for mode in ("columns", "sampler"):
            ops.DCN_BWD = mode
            xs, os_, ms, ws = (t.clone().requires_grad_(True) for t in (x, off, mask, wt))
            out = AG.modulated_deform_conv2d(xs, os_, ms, ws, None, 1, 1, 1, 1, 8)
            for _ in range(3):
                torch.autograd.grad((out * dy).sum(), [xs, os_, ms, ws], retain_graph=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                torch.autograd.grad((out * dy).sum(), [xs, os_, ms, ws], retain_graph=True)
            e1.record(); torch.cuda.synchronize()
            print(f"   backward via {mode}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per call (eager, incl. launches)")
