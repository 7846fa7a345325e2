#!/usr/bin/env python3
"""conv3x3 64->64 time vs number of tiles (blocks per CU), to separate per-block latency from sharing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from eavsr_amd import ops
dev = torch.device("cuda:0")
w33, b = torch.randn(64, 64, 3, 3, device=dev) * 0.05, torch.randn(64, device=dev) * 0.1
def T(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
for (n, h, w) in [(1, 16, 32), (1, 64, 64), (1, 128, 256), (1, 180, 320), (2, 180, 320), (4, 176, 320), (4, 180, 320), (8, 180, 320), (16, 180, 320), (4, 720, 1280)]:
    x = torch.randn(n, 64, h, w, device=dev)
    tiles = n * ((h + 15) // 16) * ((w + 31) // 32)
    dt = T(lambda: ops.conv2d(x, w33, b, act="relu"), reps=5 if h > 500 else 20)
    fl = 2 * 64 * 64 * 9 * n * h * w
    print(f"n={n:2d} {h}x{w}: tiles {tiles:5d} ({tiles/256:.2f}/CU)  {dt*1e6:9.1f} us  {fl/dt/1e12:6.1f} TF", flush=True)
