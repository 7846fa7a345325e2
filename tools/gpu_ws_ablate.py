#!/usr/bin/env python3
"""Timing ablations of the wave-specialised DCNv2 kernel: diagnostic libraries built by tools/build_ws_diag.sh (results wrong)."""
import ctypes as C
import glob
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eavsr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n, h, w = int(os.environ.get("N", 2)), 180, 320
sigma = float(os.environ.get("SIGMA", 1.5))
x = torch.randn(n, 64, h, w, device=dev)
xil = ops.to_il8(x)
off = torch.randn(n, 144, h, w, device=dev) * sigma
mask = torch.rand(n, 72, h, w, device=dev)
wt = torch.randn(64, 64, 3, 3, device=dev) * 0.05
b = torch.randn(64, device=dev) * 0.1
wx = ops._packed_dcn_x9(wt)
out = torch.empty(n, 64, h, w, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
for path in sorted(glob.glob(os.path.join(ROOT, "eavsr_amd", "lib", "libws_*.so"))):
    lib = C.CDLL(path)
    res = []
    for nprod in (6, 9):
        call = lambda: lib.eavsr_dcnv2_ws_f32(p(xil), p(off), p(mask), p(wx), p(b), p(out), n, 64, h, w, 64, 8, nprod, 0, None)
        for _ in range(3):
            assert call() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10 * 1000)
    print(f"{os.path.basename(path):28s} x6 {res[0]:8.1f} us   x9 {res[1]:8.1f} us", flush=True)
