"""3x3 64 -> 64 convolution at small launches: eavsr_conv3x3_f32x6s (exact bf16x6) against the fp32-MFMA small kernel, every
epilogue RCABlock's forward / backward uses.  HIP events around REPS launches on one stream (serialised), rotation of the variants.
  python tools/gpu_conv3_small_time.py [n h w]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from eavsr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
shape = [int(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else [2, 96, 96]
n, h, w = shape
REPS = int(os.environ.get("REPS", "200"))
torch.manual_seed(0)
x = torch.randn(n, 64, h, w, device=dev)
wt = torch.randn(64, 64, 3, 3, device=dev) / 24
b = torch.randn(64, device=dev) * 0.1
r = torch.relu(torch.randn(n, 64, h, w, device=dev))
kinds = {"relu": dict(act="relu"), "sums": dict(chan_partial=True), "residual": dict(residual=r), "relu_mask": dict(act="relu_mask", residual=r)}


def run(mode, kw):
    ops.CONV3_SMALL = mode
    for _ in range(5):
        ops.conv2d(x, wt, b, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        ops.conv2d(x, wt, b, **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REPS


print(f"shape {n} x 64 x {h} x {w}, {REPS} launches per figure (us per launch incl. the enqueue gap of an eager loop)")
for rot in range(3):
    for name, kw in kinds.items():
        t = {m: run(m, kw) for m in ("x6s", "direct")}
        print(f"  rot {rot} {name:10s} x6s {t['x6s']:7.2f}   direct {t['direct']:7.2f}")
# serialised in one graph: no enqueue gaps
for mode in ("x6s", "direct"):
    ops.CONV3_SMALL = mode
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ops.conv2d(x, wt, b, act="relu")
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            y = x
            for _ in range(100):
                y = ops.conv2d(y, wt, b, act="relu")
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(10):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
        print(f"  graph of 100 dependent relu convolutions: {mode:7s} {e0.elapsed_time(e1) * 1e3 / 1000:7.2f} us per launch")
