#!/usr/bin/env python3
"""debug: where does the grouped kernel's scaled-residual epilogue differ from the round-5 library's?"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from eavsr_amd import ops, _native as N
dev = torch.device("cuda:0")
old = C.CDLL(os.path.join(ROOT, "eavsr_amd", "lib", "libeavsr_r5epi.so"))
old.eavsr_conv3x3_wino4_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
torch.manual_seed(0)
n, h, w = 2, 180, 320
x = torch.randn(n, 64, h, w, device=dev); wt = torch.randn(64, 64, 3, 3, device=dev) * 0.05; b = torch.randn(64, device=dev) * 0.1
res = torch.randn(n, 64, h, w, device=dev); rs = torch.rand(n, 64, device=dev)
wu = ops._packed_wino([wt], four=True)
def run(lib, with_rs=True, act=0):
    out = torch.full((n, 64, h, w), float("nan"), device=dev)
    d = N.ConvDesc()
    d.src[0] = x.data_ptr(); d.src_c[0] = 64; d.n_src = 1; d.ksize = 3
    d.bias = b.data_ptr(); d.out = out.data_ptr(); d.residual = res.data_ptr()
    d.res_scale = rs.data_ptr() if with_rs else None
    d.n, d.h, d.w, d.cin, d.cout = n, h, w, 64, 64
    d.act = act; d.slope = 0.1
    rc = lib.eavsr_conv3x3_wino4_f32(C.byref(d), C.c_void_p(wu.data_ptr()), None)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return out
import glob
libs = [("product", ops.lib())]
for pth in sorted(glob.glob(os.path.join(ROOT, "eavsr_amd", "lib", "libw6dbg_*.so"))):
    l_ = C.CDLL(pth)
    l_.eavsr_conv3x3_wino4_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    libs.append((os.path.basename(pth), l_))
for name, lib_ in libs:
    ref = run(old, True, 0)
    cnt = []
    for k in range(4):
        y = run(lib_, True, 0)
        cnt.append(int(((y != ref) | torch.isnan(y)).sum()))
    print(f"{name:20s} rsc mismatches per run: {cnt}")
for act in (0,):
    ref = run(old, True, act)
    for k in range(1):
        y = run(ops.lib(), True, act)
        bad = (y != ref) | torch.isnan(y)
        print(f"act {act} run {k}: mismatches {int(bad.sum())} of {y.numel()}, max diff {(y - ref).abs().nan_to_num(1e9).max().item():.3e}, nan {int(torch.isnan(y).sum())}")
        if bad.any():
            idx = bad.nonzero()
            print("  samples", sorted(set(idx[:, 0].tolist())), "channels", sorted(set(idx[:, 1].tolist()))[:70])
            print("  rows", sorted(set(idx[:, 2].tolist()))[:40], "... cols", sorted(set(idx[:, 3].tolist()))[:40])
            i = idx[0].tolist(); print("  first", i, y[tuple(i)].item(), ref[tuple(i)].item(), "res", res[tuple(i)].item(), "scale", rs[i[0], i[1]].item())
            # is the result what another scale / residual row would give?
            conv = (ref - res) / rs[:, :, None, None]
            for dc in (-3, -2, -1, 1, 2, 3):
                c2 = min(63, max(0, i[1] + dc))
                print(f"    with scale of channel {c2}: {(conv[tuple(i)] * rs[i[0], c2] + res[tuple(i)]).item():.6f}")
    y2 = run(ops.lib(), False, act); r2 = run(old, False, act)
    print(f"act {act} plain residual: equal {bool(torch.equal(y2, r2))}")
