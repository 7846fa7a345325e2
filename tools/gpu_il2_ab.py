#!/usr/bin/env python3
"""A/B timing of eavsr_dcnv2_il2_f32 builds (eavsr_amd/lib/libil2_*.so, tools/build_il2_diag.sh) and of the product library's
il / il2 kernels: every library is measured ROUNDS times in rotation (so that clock ramps and drifts hit all alike); a sample
is the mean of INNER back-to-back launches; reported: the median over rounds of the per-round median."""
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
sigma = float(os.environ.get("SIGMA", 0.5))
reps, inner, rounds = int(os.environ.get("REPS", 9)), int(os.environ.get("INNER", 10)), int(os.environ.get("ROUNDS", 4))
torch.manual_seed(0)
x = torch.randn(n, 64, h, w, device=dev)
xil = ops.to_il8(x)
off = torch.randn(n, 144, h, w, device=dev) * sigma
mask = torch.rand(n, 72, h, w, device=dev)
wt = torch.randn(64, 64, 3, 3, device=dev) * 0.05
b = torch.randn(64, device=dev) * 0.1
wx2, wx9 = ops._packed_dcn_il2(wt), ops._packed_dcn_x9(wt)
out = torch.empty(n, 64, h, w, device=dev)
heads = torch.cat([torch.randn(n, 32, h, w, device=dev) * 0.25 + torch.tensor([1.0, 0, 0, 1.0], device=dev).repeat(8).view(1, 32, 1, 1),
                   torch.randn(n, 16, h, w, device=dev) * sigma, torch.randn(n, 72, h, w, device=dev)], 1)
p = lambda t: C.c_void_p(t.data_ptr())
paths = sorted(glob.glob(os.path.join(ROOT, "eavsr_amd", "lib", "libil2_*.so")))
only = os.environ.get("ONLY")
if only:
    paths = [q for q in paths if any(k in os.path.basename(q) for k in only.split(","))]
plib = C.CDLL(os.path.join(ROOT, "eavsr_amd", "lib", "libeavsr_hip.so"))
cands = [("product il (round 2)", lambda hm: plib.eavsr_dcnv2_il_f32(p(xil), p(heads if hm else off), p(mask), p(wx9), p(b), p(out), n, 64, h, w, 64, 8, 6, hm, None)),
         ("product il2", lambda hm: plib.eavsr_dcnv2_il2_f32(p(xil), p(heads if hm else off), p(mask), p(wx2), p(b), p(out), n, 64, h, w, 64, 8, 6, hm, None))]
for path in paths:
    lib = C.CDLL(path)
    cands.append((os.path.basename(path), (lambda L: lambda hm: L.eavsr_dcnv2_il2_f32(p(xil), p(heads if hm else off), p(mask), p(wx2), p(b), p(out), n, 64, h, w, 64, 8, 6, hm, None))(lib)))


def med(call):
    for _ in range(3):
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


res = {name: ([], []) for name, _ in cands}
med(lambda: cands[1][1](1))      # burn the clock ramp
# every build must give the product il2 kernel's bits (timing-only EXP builds excepted: they say so)
for hm in (1, 0):
    out.zero_()
    cands[1][1](hm)
    torch.cuda.synchronize()
    want = out.clone()
    for name, fn in cands[2:]:
        out.zero_()
        fn(hm)
        torch.cuda.synchronize()
        print(f"{name:28s} heads={hm} bits {'same' if torch.equal(out, want) else 'DIFFER max %.3g' % (out - want).abs().max().item()}")
for r in range(rounds):
    for name, fn in cands:
        res[name][0].append(med(lambda: fn(1)))
        res[name][1].append(med(lambda: fn(0)))
m = lambda v: sorted(v)[len(v) // 2]
print(f"n={n} sigma={sigma} rounds={rounds}: median (min..max) us")
for name, _ in cands:
    hh, ee = res[name]
    print(f"{name:28s} heads x6 {m(hh):7.1f} ({min(hh):.1f}..{max(hh):.1f})   explicit x6 {m(ee):7.1f} ({min(ee):.1f}..{max(ee):.1f})", flush=True)
