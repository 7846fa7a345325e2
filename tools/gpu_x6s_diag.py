#!/usr/bin/env python3
"""eavsr_conv3x3_f32x6s variants (tools/build_x6s_diag.sh): time per launch at a training crop, equality with the shipped kernel
for libx6s_v_*, and the phase stamps of libx6s_stamps (clock ticks of wave 0, averaged over workgroups)."""
import ctypes as C
import glob
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eavsr_amd import _native as NAT  # noqa: E402

dev = torch.device("cuda:0")
n, h, w = [int(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else (2, 96, 96)
REPS = int(os.environ.get("REPS", "200"))
p = lambda t: C.c_void_p(t.data_ptr())
paths = sorted(glob.glob(os.path.join(ROOT, "eavsr_amd", "lib", "libx6s_*.so")))
paths.sort(key=lambda q: (not q.endswith("libx6s_full.so"), q))
torch.manual_seed(1)
x = torch.randn(n, 64, h, w, device=dev)
wt = torch.randn(64, 64, 3, 3, device=dev) / 24
b = torch.randn(64, device=dev) * 0.1
ref = None
libs = []
for path in paths:
    lib = C.CDLL(path)
    lib.eavsr_conv_weight_x6_bytes.restype = C.c_size_t
    wp = torch.empty(lib.eavsr_conv_weight_x6_bytes(3, 64, 64), device=dev, dtype=torch.uint8)
    assert lib.eavsr_pack_conv_weight_x6(p(wt), p(wp), 3, 64, 64, None) == 0
    out = torch.zeros(n, 64, h, w, device=dev)
    d = NAT.ConvDesc()
    d.src[0] = p(x); d.src_c[0] = 64; d.n_src = 1; d.ksize = 3; d.bias = p(b); d.out = p(out)
    d.n, d.h, d.w, d.cin, d.cout = n, h, w, 64, 64
    d.act = 1
    libs.append((os.path.basename(path), lib, wp, out, d))


def call(ent):
    return ent[1].eavsr_conv3x3_f32x6s(C.byref(ent[4]), p(ent[2]), None)


for ent in libs:
    assert call(ent) == 0
    torch.cuda.synchronize()
    if ent[0] == "libx6s_full.so":
        ref = ent[3].clone()
    elif ent[0].startswith("libx6s_v_"):
        dd = (ent[3] - ref).abs().max().item()
        print(f"{ent[0]}: max |diff| vs full {dd:.2e}")
for rot in range(3):
    line = f"rot {rot}:"
    for ent in libs:
        for _ in range(5):
            call(ent)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            call(ent)
        e1.record()
        torch.cuda.synchronize()
        line += f"  {ent[0][7:-3]} {e0.elapsed_time(e1) * 1e3 / REPS:6.2f}"
    print(line, flush=True)
for ent in libs:
    if "stamps" in ent[0]:
        buf = (C.c_ulonglong * 8)()
        ent[1].eavsr_debug_x6s_stamps(buf, 1)
        call(ent)
        torch.cuda.synchronize()
        ent[1].eavsr_debug_x6s_stamps(buf, 1)
        nwg = n * ((h + 7) // 8) * ((w + 31) // 32) * 2
        names = ["requests issued", "loads landed + chunk 0 split", "first barrier", "k-steps", "epilogue issued", "stores acknowledged"]
        tot = sum(buf[i] for i in range(6))
        print(f"{ent[0]}: " + "  ".join(f"{names[i]} {buf[i] / nwg:.0f}" for i in range(6)) + f"  total {tot / nwg:.0f} ticks per workgroup (s_memtime: 100 MHz)")

# in-step behaviour: a HIP graph of 120 dependent convolutions with 60 DIFFERENT weights (each launch's weight is cold in L2)
NW = 60
wts = [torch.randn(64, 64, 3, 3, device=dev) / 24 for _ in range(NW)]
for ent in libs:
    name, lib = ent[0], ent[1]
    wps = []
    for wt_ in wts:
        wp_ = torch.empty(lib.eavsr_conv_weight_x6_bytes(3, 64, 64), device=dev, dtype=torch.uint8)
        assert lib.eavsr_pack_conv_weight_x6(p(wt_), p(wp_), 3, 64, 64, None) == 0
        wps.append(wp_)
    bufs = [torch.randn(n, 64, h, w, device=dev), torch.empty(n, 64, h, w, device=dev)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for k in range(120):
                d = NAT.ConvDesc()
                d.src[0] = p(bufs[k & 1]); d.src_c[0] = 64; d.n_src = 1; d.ksize = 3; d.bias = p(b); d.out = p(bufs[(k + 1) & 1])
                d.n, d.h, d.w, d.cin, d.cout = n, h, w, 64, 64
                d.act = 1
                assert lib.eavsr_conv3x3_f32x6s(C.byref(d), p(wps[k % NW]), C.c_void_p(s.cuda_stream)) == 0
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(10):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
        print(f"{name}: graph of 120 dependent convolutions, 60 weights: {e0.elapsed_time(e1) * 1e3 / 1200:6.2f} us per launch", flush=True)
