#!/usr/bin/env python3
"""Where a conv_wino6_kernel<3> launch spends its time: per-workgroup wall-clock stamps (libwino4_*tl*.so built with
-DEAVSR_W4_TIMELINE by tools/build_wino4_diag.sh) and back-to-back timings of every libwino4_*.so variant, each checked
against the base library's output."""
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
h, w = 180, 320
for n in [int(v) for v in os.environ.get("NS", "2,4").split(",")]:
    print(f"==== N = {n}")
    torch.manual_seed(0)
    x = torch.randn(n, 64, h, w, device=dev)
    wt = torch.randn(64, 64, 3, 3, device=dev) * 0.05
    b = torch.randn(64, device=dev) * 0.1
    wu = ops._packed_wino([wt], four=True)
    out = torch.empty(n, 64, h, w, device=dev)
    part = torch.empty(n * 23 * 5 * 64, device=dev)
    d = N.ConvDesc()
    d.src[0] = x.data_ptr(); d.src_c[0] = 64; d.n_src = 1; d.ksize = 3
    d.bias = b.data_ptr(); d.out = out.data_ptr()
    d.n, d.h, d.w, d.cin, d.cout = n, h, w, 64, 64
    d.act = 1
    ref_out = None
    paths = sorted(glob.glob(os.path.join(ROOT, "eavsr_amd", "lib", "libwino4_*.so")), key=lambda p: (0 if "base" in p else 1, p))
    for path in paths:
        lib = C.CDLL(path)
        lib.eavsr_conv3x3_wino4_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        call = lambda: lib.eavsr_conv3x3_wino4_f32(C.byref(d), C.c_void_p(wu.data_ptr()), None)
        out.zero_()
        for _ in range(3):
            assert call() == 0, call()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                call()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1000)
        # "in-step-like": 48 weight sets and 6 inputs in rotation (28 MB of weights + 177 MB of activations between two uses
        # of the same bytes: nothing is L2-resident, everything memory-side-cache resident, as in the forward)
        if n == 2:
            if "wus" not in globals():
                globals()["wus"] = [ops._packed_wino([torch.randn(64, 64, 3, 3, device=dev) * 0.05], four=True) for _ in range(48)]
                globals()["xs"] = [torch.randn(n, 64, h, w, device=dev) for _ in range(6)]
            wus, xs = globals()["wus"], globals()["xs"]
            def call_rot(i):
                d.src[0] = xs[i % 6].data_ptr()
                return lib.eavsr_conv3x3_wino4_f32(C.byref(d), C.c_void_p(wus[i % 48].data_ptr()), None)
            for i in range(48):
                call_rot(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(96):
                call_rot(i)
            e1.record()
            torch.cuda.synchronize()
            cold = e0.elapsed_time(e1) / 96 * 1000
            d.src[0] = x.data_ptr()
            call(); torch.cuda.synchronize()
        else:
            cold = float("nan")
        note = f"   rotating weights / inputs {cold:6.1f} us"
        if ref_out is None and "base" in path:
            ref_out = out.clone()
        elif ref_out is not None:
            note += f"   max |out - base| = {(out - ref_out).abs().max().item():.3e}"
        print(f"{os.path.basename(path):36s} {min(ts):7.1f} us (median {sorted(ts)[2]:.1f}){note}")
        if hasattr(lib, "eavsr_debug_w4_timeline"):
            buf = (C.c_ulonglong * (512 * 16))()
            nwg = min(256, n * 23 * 5)
            for mode in ("back-to-back", "alone"):
                if mode == "alone":
                    torch.cuda.synchronize(); call(); torch.cuda.synchronize()
                else:
                    for _ in range(4):
                        call()
                lib.eavsr_debug_w4_timeline(buf)
                a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 8, 2)[:nwg].astype(np.int64)
                rt, cy = a[:, :6, 0], a[:, :6, 1]          # 100 MHz wall clock, shader cycles
                t0 = rt[:, 0].min()
                us = (rt - t0) / 100.0
                names = ["entry", "set-up done / DMA issued", "first DMA landed", "GEMM loop done", "epilogue issued", "stores acked"]
                print(f"  [{mode}] per-workgroup stamps, us after the first workgroup's entry: median (min .. max)")
                for i, nm in enumerate(names):
                    print(f"    {nm:26s} {np.median(us[:, i]):7.2f} ({us[:, i].min():6.2f} .. {us[:, i].max():6.2f})")
                dcy = np.diff(cy, axis=1)
                drt = np.diff(rt, axis=1)
                print("    phase lengths, shader cycles (median): " + "  ".join(f"{int(np.median(dcy[:, i]))}" for i in range(5)))
                tot_cy = (cy[:, 5] - cy[:, 0]); tot_rt = (rt[:, 5] - rt[:, 0])
                print(f"    clock: {np.median(tot_cy / np.maximum(tot_rt, 1)) * 100:.0f} MHz;  entry spread {us[:, 0].max():.2f} us;  last ack {us[:, 5].max():.2f} us")
