#!/usr/bin/env python3
"""Anatomy of a steady-state iteration of conv_wino6_kernel<3>: libwino4_itstamp*.so (-DEAVSR_W4_ITSTAMP) stamps every wave at
eight points of iterations 8..11; printed as cycles after the iteration's first barrier release, median over workgroups."""
import ctypes as C
import glob
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from eavsr_amd import ops, _native as N  # noqa: E402

dev = torch.device("cuda:0")
n, h, w = int(os.environ.get("N", 2)), 180, 320
torch.manual_seed(0)
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
names_lock = ["top", "own DMA landed", "barrier released", "early DMA issued", "transform done", "GEMM half", "late DMA issued", "GEMM done"]
names_pp = ["top", "B: barrier", "GEMM done", "own DMA landed", "barrier", "service done", "bottom (A: barrier)", "-"]
for path in sorted(glob.glob(os.path.join(ROOT, "eavsr_amd", "lib", "libwino4_itstamp*.so"))):
    lib = C.CDLL(path)
    lib.eavsr_conv3x3_wino4_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    call = lambda: lib.eavsr_conv3x3_wino4_f32(C.byref(d), C.c_void_p(wu.data_ptr()), None)
    for _ in range(5):
        assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record(); torch.cuda.synchronize()
    print(f"{os.path.basename(path)}: {e0.elapsed_time(e1) / 20 * 1000:.1f} us per launch")
    buf = (C.c_ulonglong * (256 * 256))()
    lib.eavsr_debug_w4_itstamps(buf)
    nwg = min(256, n * 23 * 5)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 4, 8, 8)[:nwg].astype(np.int64)   # [wg][iteration 8..11][wave][point]
    pp = "_pp" in os.path.basename(path)
    names = names_pp if pp else names_lock
    ref = 0 if pp else 2
    for itx in range(4):
        rel = a[:, itx] - a[:, itx, :4, ref].min(axis=1)[:, None, None]     # cycles after the first wave of group A was at the reference point
        nxt = (a[:, itx + 1, :4, ref].min(axis=1) - a[:, itx, :4, ref].min(axis=1)) if itx < 3 else None
        print(f"  iteration {8 + itx}" + ("" if pp else f" (duty pair: waves {2 * (itx & 3)}, {2 * (itx & 3) + 1})")
              + (f": {int(np.median(nxt))} cycles to the same point of the next iteration" if nxt is not None else ""))
        print("    wave  " + "  ".join(f"{nm:>17s}" for nm in names))
        for wv in range(8):
            print(f"    {wv:4d}  " + "  ".join(f"{int(np.median(rel[:, wv, i])):17d}" for i in range(8)))
