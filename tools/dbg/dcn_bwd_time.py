#!/usr/bin/env python3
"""time eavsr_dcnv2_bwd_f32 of every eavsr_amd/lib/libdcnb_*.so (ablation builds of csrc/dcn_bwd.hip + capi.hip) at 2 x 64 x 96 x 96"""
import ctypes as C, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from eavsr_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
n, h, w = 2, 96, 96
x = torch.randn(n, 64, h, w, device=dev); off = torch.randn(n, 144, h, w, device=dev); mask = torch.rand(n, 72, h, w, device=dev)
wt = torch.randn(64, 64, 3, 3, device=dev) / 24; dy = torch.randn(n, 64, h, w, device=dev)
xil = ops.to_il8(x)
for pth in sorted(glob.glob(os.path.join(ROOT, "eavsr_amd", "lib", "libdcnb_*.so"))):
    lib = C.CDLL(pth)
    lib.eavsr_dcnv2_bwd_workspace_floats.restype = C.c_int64
    ws = torch.empty(int(lib.eavsr_dcnv2_bwd_workspace_floats(n, h, w)), device=dev)
    dxil = torch.zeros_like(xil); doff = torch.empty_like(off); dm = torch.empty_like(mask); dw = torch.empty_like(wt)
    P = lambda t: C.c_void_p(t.data_ptr())
    call = lambda: lib.eavsr_dcnv2_bwd_f32(P(xil), P(off), P(mask), P(wt), P(dy), P(dxil), P(doff), P(dm), P(dw), P(ws), n, 64, h, w, 64, 8, 0, None)
    for _ in range(3):
        assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record(); torch.cuda.synchronize()
    print(f"{os.path.basename(pth):28s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us per call (pack + data + weight + reduce)")
