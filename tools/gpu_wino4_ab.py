#!/usr/bin/env python3
"""One library (EAVSR_LIB_PATH, default the product library) on the 3x3 64 -> 64 F(4x4,3x3) convolution, the dominant kernel:
median of REPS samples of INNER back-to-back launches at 2 and 4 clips, and a hash of the output bits (so that builds can be
compared for bit-identity).  tools/visits/r4_z.sh runs the builds in rotation, one process each."""
import hashlib
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eavsr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
h, w = 180, 320
reps, inner = int(os.environ.get("REPS", 15)), int(os.environ.get("INNER", 10))
torch.manual_seed(0)
wt = torch.randn(64, 64, 3, 3, device=dev) * 0.05
b = torch.randn(64, device=dev) * 0.1
out = []
# FORM: relu (default) | sums (ReLU + per-tile channel sums: conv-1 of an RCAB) | rsc (residual + res_scale: conv-2 with the tail as
# its epilogue) | res (plain residual)
form = os.environ.get("FORM", "relu")
for n in (2, 4):
    x = torch.randn(n, 64, h, w, device=dev)
    res = torch.randn(n, 64, h, w, device=dev)
    rs = torch.rand(n, 64, device=dev)
    call = {"relu": lambda: ops.conv2d(x, wt, b, act="relu"),
            "sums": lambda: ops.conv2d(x, wt, b, act="relu", chan_partial=True)[0],
            "rsc": lambda: ops.conv2d(x, wt, b, residual=res, res_scale=rs),
            "res": lambda: ops.conv2d(x, wt, b, residual=res)}[form]
    for _ in range(5):
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
    out.append(f"n={n} {ts[len(ts) // 2]:6.1f} us ({ts[0]:.1f}..{ts[-1]:.1f}) bits {hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:10]}")
print(f"{os.path.basename(os.environ.get('EAVSR_LIB_PATH', 'libeavsr_hip.so')):24s} {form:5s} " + "   ".join(out), flush=True)
