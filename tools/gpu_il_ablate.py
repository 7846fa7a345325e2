#!/usr/bin/env python3
"""Timing ablations of the IL8 DCNv2 kernel: diagnostic libraries built by tools/build_il_diag.sh (results wrong)."""
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
heads = torch.randn(n, 120, h, w, device=dev) * 0.3          # AdaptBlockOffset head channels (heads mode, dg = 8)
heads[:, :32] += torch.tensor([1.0, 0.0, 0.0, 1.0], device=dev).repeat(8).view(1, 32, 1, 1)
p = lambda t: C.c_void_p(t.data_ptr())
ref = {}
paths = sorted(glob.glob(os.path.join(ROOT, "eavsr_amd", "lib", "libil_*.so")))
paths.sort(key=lambda q: (not q.endswith("libil_full.so"), q))      # the reference first
for path in paths:
    lib = C.CDLL(path)
    res = []
    exact = os.path.basename(path).startswith(("libil_full", "libil_v_")) and "stamps" not in path    # variants that must reproduce the full kernel
    for hm in (0, 1):
        for nprod in (6, 9):
            out.zero_()
            assert lib.eavsr_dcnv2_il_f32(p(xil), p(heads if hm else off), p(mask), p(wx), p(b), p(out), n, 64, h, w, 64, 8, nprod, hm, None) == 0
            torch.cuda.synchronize()
            if path.endswith("libil_full.so"):
                ref[(hm, nprod)] = out.clone()
            elif exact:
                d = (out - ref[(hm, nprod)]).abs().max().item()
                print(f"    {os.path.basename(path)} heads={hm} x{nprod}: max |out - full| = {d:.3g}", flush=True)
    callh = lambda: lib.eavsr_dcnv2_il_f32(p(xil), p(heads), p(mask), p(wx), p(b), p(out), n, 64, h, w, 64, 8, 6, 1, None)
    for _ in range(3):
        callh()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        callh()
    e1.record()
    torch.cuda.synchronize()
    heads_us = e0.elapsed_time(e1) / 10 * 1000
    for nprod in (6, 9):
        call = lambda: lib.eavsr_dcnv2_il_f32(p(xil), p(off), p(mask), p(wx), p(b), p(out), n, 64, h, w, 64, 8, nprod, 0, None)
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
    print(f"{os.path.basename(path):28s} x6 {res[0]:8.1f} us   x9 {res[1]:8.1f} us   heads x6 {heads_us:8.1f} us", flush=True)
    if "stamps" in path:
        buf = (C.c_ulonglong * 16)()
        lib.eavsr_debug_il_stamps(buf, 1)
        lib.eavsr_dcnv2_il_f32(p(xil), p(off), p(mask), p(wx), p(b), p(out), n, 64, h, w, 64, 8, 6, 0, None)
        lib.eavsr_debug_il_stamps(buf, 1)
        names = ["loop/bookkeeping", "wait vmcnt(0)", "barrier", "store", "dma+params issue", "prologue", "pipeline", "fixup+tail"]
        for wv in (0, 1):
            tot = sum(buf[wv * 8 + i] for i in range(8))
            print(f"  wave {wv * 4}: " + "  ".join(f"{names[i]} {100.0 * buf[wv * 8 + i] / max(tot, 1):.1f}%" for i in range(8)) + f"  (total {tot / 256:.0f} cycles per workgroup)")
