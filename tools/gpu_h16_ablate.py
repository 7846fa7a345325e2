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
h, w = int(os.environ.get("H", 180)), int(os.environ.get("W", 320))
n_list = tuple(int(v) for v in os.environ.get("N", "2,4").split(","))
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
for path in sorted(glob.glob(os.path.join(ROOT, "eavsr_amd", "lib", "libh16_*.so"))):
    lib = C.CDLL(path)
    res = []
    for n in n_list:
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
    print(f"{os.path.basename(path):24s} " + "   ".join(f"n={n} {r:7.1f} us" for n, r in zip(n_list, res)), flush=True)
    if "stamps" in path:
        buf = (C.c_ulonglong * 8)()
        lib.eavsr_debug_h16_stamps(buf, 1)
        call()
        lib.eavsr_debug_h16_stamps(buf, 1)
        names = ["wait DMA + barrier", "reads + MFMAs", "border clear + DMA issue", "bias/round/stage", "channel sums", "stores"]
        tot = sum(buf[i] for i in range(6))
        print(f"  waves 0 + 4 (n={n_list[-1]}): " + "  ".join(f"{names[i]} {100.0 * buf[i] / max(tot, 1):.1f}%" for i in range(6)) + f"  (total {tot / 512:.0f} cycles per wave)")
        tiles_g = ((h + 7) // 8) * ((w + 31) // 32) * n_list[-1] / 512.0      # tiles per wave group
        print("    cycles per tile of a group: " + "  ".join(f"{names[i]} {buf[i] / 512 / tiles_g:.0f}" for i in range(6)) + f"  | sum {tot / 512 / tiles_g:.0f}")
