#!/usr/bin/env python3
"""Timing ablations of the Winograd conv kernel: diagnostic libraries eavsr_amd/lib/libwino4_*.so built from
conv_wino6.hip + capi.hip (tools/build_wino4_diag.sh) with -DEAVSR_WINO_EXP_* (results wrong)."""
import ctypes as C
import glob
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eavsr_amd import ops, _native as N  # noqa: E402

dev = torch.device("cuda:0")
n, h, w = int(os.environ.get("N", 2)), 180, 320
x = torch.randn(n, 64, h, w, device=dev)
wt = torch.randn(64, 64, 3, 3, device=dev) * 0.05
b = torch.randn(64, device=dev) * 0.1
wu = ops._packed_wino([wt], four=True)
out = torch.empty(n, 64, h, w, device=dev)
d = N.ConvDesc()
d.src[0] = x.data_ptr(); d.src_c[0] = 64; d.n_src = 1; d.ksize = 3
d.bias = b.data_ptr(); d.out = out.data_ptr()
d.n, d.h, d.w, d.cin, d.cout = n, h, w, 64, 64
d.act = 1
for path in sorted(glob.glob(os.path.join(ROOT, "eavsr_amd", "lib", "libwino4_*.so"))):
    lib = C.CDLL(path)
    lib.eavsr_conv3x3_wino4_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    call = lambda: lib.eavsr_conv3x3_wino4_f32(C.byref(d), C.c_void_p(wu.data_ptr()), None)
    for _ in range(3):
        assert call() == 0, call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    note = ""
    if "base" in path:
        ref_out = out.clone()
    elif "uregs" in path and "ref_out" in globals():
        note = f"   max |out - base| = {(out - ref_out).abs().max().item():.3e}"
    print(f"{os.path.basename(path):32s} {e0.elapsed_time(e1) / 20 * 1000:8.1f} us{note}")
    if "stamps" in path:
        buf = (C.c_ulonglong * 32)()
        lib.eavsr_debug_w4_stamps(buf, 1)
        call()
        lib.eavsr_debug_w4_stamps(buf, 1)
        names = ["prologue", "wait own DMA", "barrier", "DMA issue", "transform", "GEMM", "epilogue", "-"]
        nwg = min(256, n * 23 * 5)
        for wv in range(4):
            tot = sum(buf[wv * 8 + i] for i in range(8))
            print(f"  wave {2 * wv}: " + "  ".join(f"{names[i]} {100.0 * buf[wv * 8 + i] / max(tot, 1):.1f}%" for i in range(7)) + f"  (total {tot / nwg:.0f} cycles per workgroup)")
