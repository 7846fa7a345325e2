#!/usr/bin/env python3
"""flow_warp_pair alone: time against shape, IL8 form (fp32 / bf16 / none) and flow magnitude (what bounds the paired warp?)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eavsr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
for (n, h, w) in ((2, 180, 320), (4, 256, 256), (1, 540, 960)):
    xa, xb = torch.randn(n, 64, h, w, device=dev), torch.randn(n, 64, h, w, device=dev)
    for sigma in (0.0, 0.5):
        flow = torch.randn(n, 2, h, w, device=dev) * sigma
        for il8 in (False, True, "bf16", "fp16"):
            call = lambda: ops.flow_warp_pair(xa, xb, flow, b_il8=il8)
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                call()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            px = n * h * w
            by = px * (64 * 4 * 2 + 64 * 4 + (64 * 2 if isinstance(il8, str) else 64 * 4) + 8)
            print(f"{n}x64x{h}x{w} sigma {sigma} il8={il8!s:5s} {us:7.1f} us  {by / us / 1e6:5.2f} TB/s", flush=True)
