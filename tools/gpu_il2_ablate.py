#!/usr/bin/env python3
"""Timing ablations of eavsr_dcnv2_il2_f32: diagnostic libraries built by tools/build_il2_diag.sh (results wrong by
construction except libil2_full / libil2_stamps / libil2_v_*)."""
import ctypes as C
import glob
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eavsr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n, h, w = int(os.environ.get("N", 2)), 180, 320
sigma = float(os.environ.get("SIGMA", 1.5))
reps = int(os.environ.get("REPS", 15))
torch.manual_seed(0)
x = torch.randn(n, 64, h, w, device=dev)
xil = ops.to_il8(x)
off = torch.randn(n, 144, h, w, device=dev) * sigma
mask = torch.rand(n, 72, h, w, device=dev)
wt = torch.randn(64, 64, 3, 3, device=dev) * 0.05
b = torch.randn(64, device=dev) * 0.1
wx = ops._packed_dcn_il2(wt)
out = torch.empty(n, 64, h, w, device=dev)
heads = torch.cat([torch.randn(n, 32, h, w, device=dev) * 0.25 + torch.tensor([1.0, 0, 0, 1.0], device=dev).repeat(8).view(1, 32, 1, 1),
                   torch.randn(n, 16, h, w, device=dev) * sigma, torch.randn(n, 72, h, w, device=dev)], 1)
p = lambda t: C.c_void_p(t.data_ptr())
ref = {}
paths = sorted(glob.glob(os.path.join(ROOT, "eavsr_amd", "lib", "libil2_*.so")))
only = os.environ.get("ONLY")
if only:
    paths = [q for q in paths if any(k in os.path.basename(q) for k in only.split(",")) or q.endswith("libil2_full.so")]
paths.sort(key=lambda q: (not q.endswith("libil2_full.so"), q))      # the reference first


def med(call, inner=10):
    """median over `reps` samples of the mean duration of `inner` back-to-back launches (host gaps amortised)"""
    for _ in range(5):
        assert call() == 0
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            call()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    ts.sort()
    return ts[len(ts) // 2]


# the first library measured reads ~5 us high (clock ramp): burn the ramp on the reference before any number is taken
if paths:
    _lib0 = C.CDLL(paths[0])
    med(lambda: _lib0.eavsr_dcnv2_il2_f32(p(xil), p(heads), p(mask), p(wx), p(b), p(out), n, 64, h, w, 64, 8, 6, 1, None), inner=20)

# the round-2 kernel (eavsr_dcnv2_il_f32 of the product library), timed the same way
_plib = C.CDLL(os.path.join(ROOT, "eavsr_amd", "lib", "libeavsr_hip.so"))
_wx9 = ops._packed_dcn_x9(wt)
for _ in range(2):
    t_h = med(lambda: _plib.eavsr_dcnv2_il_f32(p(xil), p(heads), p(mask), p(_wx9), p(b), p(out), n, 64, h, w, 64, 8, 6, 1, None))
    t_e = med(lambda: _plib.eavsr_dcnv2_il_f32(p(xil), p(off), p(mask), p(_wx9), p(b), p(out), n, 64, h, w, 64, 8, 6, 0, None))
    t_2 = med(lambda: _plib.eavsr_dcnv2_il2_f32(p(xil), p(heads), p(mask), p(wx), p(b), p(out), n, 64, h, w, 64, 8, 6, 1, None))
    print(f"{'round-2 il (product lib)':28s} heads x6 {t_h:8.1f} us   explicit x6 {t_e:8.1f} us   | product il2 heads x6 {t_2:8.1f} us", flush=True)

for path in paths:
    lib = C.CDLL(path)
    base = os.path.basename(path)
    exact = base.startswith(("libil2_full", "libil2_v_", "libil2_stamps"))
    for hm in (0, 1):
        out.zero_()
        assert lib.eavsr_dcnv2_il2_f32(p(xil), p(heads if hm else off), p(mask), p(wx), p(b), p(out), n, 64, h, w, 64, 8, 6, hm, None) == 0
        torch.cuda.synchronize()
        if path.endswith("libil2_full.so"):
            ref[hm] = out.clone()
        elif exact:
            d = (out - ref[hm]).abs().max().item()
            if d != 0.0:
                print(f"    {base} heads={hm}: max |out - full| = {d:.3g}", flush=True)
    t_h = med(lambda: lib.eavsr_dcnv2_il2_f32(p(xil), p(heads), p(mask), p(wx), p(b), p(out), n, 64, h, w, 64, 8, 6, 1, None))
    t_e = med(lambda: lib.eavsr_dcnv2_il2_f32(p(xil), p(off), p(mask), p(wx), p(b), p(out), n, 64, h, w, 64, 8, 6, 0, None))
    t_9 = med(lambda: lib.eavsr_dcnv2_il2_f32(p(xil), p(heads), p(mask), p(wx), p(b), p(out), n, 64, h, w, 64, 8, 9, 1, None))
    print(f"{base:28s} heads x6 {t_h:8.1f} us   explicit x6 {t_e:8.1f} us   heads x9 {t_9:8.1f} us", flush=True)
    if "stamps" in path:
        buf = (C.c_ulonglong * 32)()
        lib.eavsr_debug_il2_stamps(buf, 1)
        lib.eavsr_dcnv2_il2_f32(p(xil), p(heads), p(mask), p(wx), p(b), p(out), n, 64, h, w, 64, 8, 6, 1, None)
        lib.eavsr_debug_il2_stamps(buf, 1)
        names = ["bookkeeping", "store", "k-steps 0-2", "vmcnt B3", "barrier B3", "k-steps 3-7", "vmcnt B8", "barrier B8", "k-step 8",
                 "fix-up", "tail", "-"]
        for wv in (0, 1):
            tot = sum(buf[wv * 16 + i] for i in range(12))
            print(f"  wave {wv * 4}: " + "  ".join(f"{names[i]} {100.0 * buf[wv * 16 + i] / max(tot, 1):.1f}%" for i in range(11))
                  + f"  (total {tot / 256:.0f} cycles per workgroup)")
