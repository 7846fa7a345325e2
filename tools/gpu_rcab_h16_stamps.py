#!/usr/bin/env python3
"""Per-phase shader cycles of eavsr_rcab_convs_h16 (librcab_stamps.so = csrc/rcab_h16.hip with -DEAVSR_RCAB_STAMPS), wave 0 and wave 4 of
every workgroup, 4 x 64 x 256 x 256 bf16 (4 tiles per workgroup)."""
import ctypes as C
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
L = C.CDLL(os.path.join(ROOT, "eavsr_amd", "lib", "librcab_stamps.so"))
call = lambda: L.eavsr_rcab_convs_h16(p(x), p(wp1), p(b1), p(wp2), p(b2), p(out), p(part), n, h, w, 2, None)
for _ in range(5):
    assert call() == 0
buf = (C.c_ulonglong * 16)()
L.eavsr_debug_rcab_stamps(buf, 1)
reps = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    call()
e1.record()
torch.cuda.synchronize()
L.eavsr_debug_rcab_stamps(buf, 0)
names = ["bookkeeping", "conv-1 k-steps", "conv-1 meetings", "intermediate write", "barrier behind it", "conv-2 k-steps", "conv-2 meetings", "last epilogue + sums"]
print(f"{e0.elapsed_time(e1) * 1e3 / reps:.1f} us per launch (with stamps)")
for wv in (0, 1):
    tot = sum(buf[wv * 8 + i] for i in range(8))
    print(f"wave {4 * wv}: total {tot / (256 * reps):.0f} cycles per workgroup and launch")
    for i in range(8):
        print(f"   {names[i]:22s} {buf[wv * 8 + i] / (256 * reps):9.0f}  {100.0 * buf[wv * 8 + i] / tot:5.1f} %")
