#!/usr/bin/env python3
"""Per-phase cycle stamps of the pipelined DCNv2 kernel (diagnostic library, tools/build_dcn_diag.sh)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eavsr_amd import ops  # noqa: E402

os.environ.setdefault("EAVSR_DCN_VARIANT", "q")
lib = C.CDLL(os.path.join(ROOT, "eavsr_amd", "lib", "libeavsr_hip_diag" + os.environ.get("SUFFIX", "") + ".so"))
dev = torch.device("cuda:0")
n, h, w = 4, 180, 320
x = torch.randn(n, 64, h, w, device=dev)
off = torch.randn(n, 144, h, w, device=dev) * float(os.environ.get("SIGMA", 1.5))
mask = torch.rand(n, 72, h, w, device=dev)
wt = torch.randn(64, 64, 3, 3, device=dev) * 0.05
b = torch.randn(64, device=dev) * 0.1
ref = ops.modulated_deform_conv2d(x, off, mask, wt, b, 1, 1, 1, 1, 8)
wp = ops.pack_cache.get([wt])
out = torch.empty_like(ref)
p = lambda t: C.c_void_p(t.data_ptr())
stamps = (C.c_ulonglong * 16)()
lib.eavsr_debug_dcn_stamps(stamps, 1)
reps = 5
for _ in range(reps):
    rc = lib.eavsr_dcnv2_f32(p(x), p(off), p(mask), p(wp), p(b), p(out), n, 64, h, w, 64, 8, None)
    assert rc == 0, rc
torch.cuda.synchronize()
print("max|diag - product| =", (out - ref).abs().max().item())
lib.eavsr_debug_dcn_stamps(stamps, 0)
blocks = (h // 4) * (w // 32) * n
names = ["contract (+loop)", "wait window + barrier", "issue DMA + offsets", "sample slots", "fix-up", "wait weights + barrier"]
for wv, base in ((0, 0), (2, 8)):
    tot = sum(stamps[base + i] for i in range(6))
    print(f"wave {wv}:")
    for i, nm in enumerate(names):
        print(f"  {nm:24s} {stamps[base + i] / (reps * blocks * 16):9.1f} cycles / chunk   {100.0 * stamps[base + i] / max(tot, 1):5.1f} %")
    print(f"  total {tot / (reps * blocks * 16):.1f} cycles / chunk")
