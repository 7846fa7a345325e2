#!/usr/bin/env python3
"""One library (EAVSR_LIB_PATH, default the product library) on the 16-bit predictor heads (5x5, 64 -> 120, `eavsr_conv5x5_c64_h16`):
median of REPS samples of INNER back-to-back launches at the shapes of configs[1]-bf16 / [2] / [4], and a hash of the output bits."""
import hashlib
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eavsr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
reps, inner = int(os.environ.get("REPS", 11)), int(os.environ.get("INNER", 10))
torch.manual_seed(0)
ws = [torch.randn(32, 64, 5, 5, device=dev) * 0.02, torch.randn(16, 64, 5, 5, device=dev) * 0.02, torch.randn(72, 64, 5, 5, device=dev) * 0.02]
bs = [torch.randn(32, device=dev) * 0.1, torch.randn(16, device=dev) * 0.1, torch.randn(72, device=dev) * 0.1]
out = []
for (n, h, w) in ((2, 180, 320), (4, 256, 256), (1, 540, 960)):
    x = ops.to_nhwc_h16(torch.randn(n, 64, h, w, device=dev), "bf16")
    call = lambda: ops.conv5x5_c64_h16(x, ws, bs)
    for _ in range(4):
        y = call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            call()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    ts.sort()
    out.append(f"{n}x{h}x{w} {ts[len(ts) // 2]:6.1f} us bits {hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:8]}")
print(f"{os.path.basename(os.environ.get('EAVSR_LIB_PATH', 'libeavsr_hip.so')):22s} " + "   ".join(out), flush=True)
