#!/usr/bin/env python3
"""Timing ablations of the 16-bit 3x3 backbone convolution: diagnostic libraries built by tools/build_h16_diag.sh (results wrong)."""
import ctypes as C
import glob
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eavsr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
h, w = 180, 320
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
for path in sorted(glob.glob(os.path.join(ROOT, "eavsr_amd", "lib", "libh16_*.so"))):
    lib = C.CDLL(path)
    res = []
    for n in (2, 4):
        x = torch.randn(n, 64, h, w, device=dev)
        xh = ops.to_nhwc_h16(x, "bf16")
        wt = torch.randn(64, 64, 3, 3, device=dev) * 0.04
        wp = ops._packed_h16(wt, 2)
        b = torch.randn(64, device=dev)
        out = torch.empty_like(xh)
        call = lambda: lib.eavsr_conv3x3_c64_h16(p(xh), p(wp), p(b), p(out), None, n, h, w, 1, 2, None)
        for _ in range(3):
            assert call() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1000)
    print(f"{os.path.basename(path):24s} n=2 {res[0]:7.1f} us   n=4 {res[1]:7.1f} us", flush=True)
    if "stamps" in path:
        buf = (C.c_ulonglong * 8)()
        lib.eavsr_debug_h16_stamps(buf, 1)
        call()
        lib.eavsr_debug_h16_stamps(buf, 1)
        names = ["wait DMA + barrier", "reads + MFMAs", "border clear + DMA issue", "bias/round/stage", "channel sums", "stores"]
        tot = sum(buf[i] for i in range(6))
        print("  wave 0 (n=4): " + "  ".join(f"{names[i]} {100.0 * buf[i] / max(tot, 1):.1f}%" for i in range(6)) + f"  (total {tot / 256:.0f} cycles per workgroup)")
