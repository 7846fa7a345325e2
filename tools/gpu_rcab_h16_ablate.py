#!/usr/bin/env python3
"""Timing ablations of eavsr_rcab_convs_h16 (eavsr_amd/lib/librcab_*.so: builds of csrc/rcab_h16.hip with -DEAVSR_RCAB_EXP_*; results wrong
by construction except `full`), 4 x 64 x 256 x 256 bf16, in rotation."""
import ctypes as C
import glob
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eavsr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
n, h, w = 4, 256, 256
x = ops.to_nhwc_h16(torch.randn(n, 64, h, w, device=dev), "bf16")
w1, w2 = torch.randn(64, 64, 3, 3, device=dev) / 24, torch.randn(64, 64, 3, 3, device=dev) / 24
b1, b2 = torch.randn(64, device=dev) * 0.1, torch.randn(64, device=dev) * 0.1
wp1, wp2 = ops._packed_h16(w1, 2), ops._packed_h16(w2, 2)
out = torch.empty_like(x)
part = torch.empty(n, 256, 64, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
libs = {os.path.basename(q)[8:-3]: C.CDLL(q) for q in sorted(glob.glob(os.path.join(ROOT, "eavsr_amd", "lib", "librcab_*.so")))}


def med(fn, reps=9, inner=10):
    for _ in range(3):
        assert fn() == 0
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    return sorted(ts)[len(ts) // 2]


res = {k: [] for k in libs}
for _ in range(3):
    for k, L in libs.items():
        res[k].append(med(lambda: L.eavsr_rcab_convs_h16(p(x), p(wp1), p(b1), p(wp2), p(b2), p(out), p(part), n, h, w, 2, None)))
for k, v in res.items():
    print(f"{k:12s} {sorted(v)[1]:7.1f} us  ({min(v):.1f}..{max(v):.1f})")
