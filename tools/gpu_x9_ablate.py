#!/usr/bin/env python3
"""Timing ablations of the bf16x9 DCNv2 kernel: diagnostic libraries built with -DEAVSR_X9_EXP_* (results wrong).

Build them first (they travel to the GPU box in eavsr_amd/lib/, which is git-ignored):
  for v in full: nooff:-DEAVSR_X9_EXP_NO_OFFSETS nodma:-DEAVSR_X9_EXP_NO_DMA nostore:-DEAVSR_X9_EXP_NO_STORE nomfma:-DEAVSR_X9_EXP_NO_MFMA; do
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -ffp-contract=fast ${v#*:} -shared \
      -o eavsr_amd/lib/libx9_${v%%:*}.so eavsr_amd/csrc/dcnv2_x9.hip eavsr_amd/csrc/capi.hip; done
Round-1 result (4 x 64 x 180 x 320, sigma 0): full 279 us, no DMA 220, no offset loads 215, no output stores 213,
no MFMAs 142 -- every component is exposed latency at 2 waves per SIMD, none is a throughput bound."""
import ctypes as C
import glob
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eavsr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n, h, w = 4, 180, 320
x = torch.randn(n, 64, h, w, device=dev)
off = torch.randn(n, 144, h, w, device=dev) * float(os.environ.get("SIGMA", 0.0))
mask = torch.rand(n, 72, h, w, device=dev)
wt = torch.randn(64, 64, 3, 3, device=dev) * 0.05
b = torch.randn(64, device=dev) * 0.1
wx = ops._packed_dcn_x9(wt)
out = torch.empty(n, 64, h, w, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
for path in sorted(glob.glob(os.path.join(ROOT, "eavsr_amd", "lib", "libx9_*.so"))):
    lib = C.CDLL(path)
    call = lambda: lib.eavsr_dcnv2_f32x9(p(x), p(off), p(mask), p(wx), p(b), p(out), n, 64, h, w, 64, 8, None)
    for _ in range(3):
        assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        call()
    e1.record()
    torch.cuda.synchronize()
    print(f"{os.path.basename(path):32s} {e0.elapsed_time(e1) / 10 * 1000:8.1f} us")
