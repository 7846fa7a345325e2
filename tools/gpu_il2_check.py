#!/usr/bin/env python3
"""Round 4: eavsr_dcnv2_il2_f32 against eavsr_dcnv2_il_f32 (same inputs; expected to agree to re-association) on the test
shapes and at the bench shape, then timings of both (HIP events on the launch stream, median of REPS)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eavsr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
reps = int(os.environ.get("REPS", 30))
torch.manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev)


def heads_of(n, D, h, w, sigma):
    return torch.cat([r(n, 4 * D, h, w) * 0.25 + torch.tensor([1.0, 0, 0, 1.0], device=dev).repeat(D).view(1, 4 * D, 1, 1),
                      r(n, 2 * D, h, w) * sigma, r(n, 9 * D, h, w)], 1)


def run(impl, *a, **k):
    ops.set_dcn_il_impl(impl)
    return ops.dcnv2_il(*a, **k)


bad = 0
for (n, c, h, w, cout, D) in [(1, 64, 24, 40, 64, 8), (2, 64, 13, 37, 64, 8), (1, 64, 10, 12, 64, 1), (1, 16, 9, 33, 32, 2),
                              (1, 64, 7, 5, 40, 8), (1, 64, 21, 68, 96, 4), (3, 64, 45, 80, 64, 8), (2, 64, 180, 320, 64, 8)]:
    for sigma in (0.0, 0.5, 2.0, 8.0):
        x = r(n, c, h, w)
        xil = ops.to_il8(x)
        wt = r(cout, c, 3, 3) / (c * 9) ** 0.5
        b = r(cout) * 0.1
        off = r(n, 18 * D, h, w) * sigma
        mask = torch.rand(n, 9 * D, h, w, device=dev)
        hd = heads_of(n, D, h, w, sigma)
        for nprod in (6, 9):
            for heads in (False, True):
                args = (xil, hd, None, wt, b, D) if heads else (xil, off, mask, wt, b, D)
                ref = run("il", *args, nprod=nprod, heads=heads)
                out = run("il2", *args, nprod=nprod, heads=heads)
                torch.cuda.synchronize()
                err = (out - ref).abs().max().item()
                sc = ref.abs().max().item()
                ok = err <= 2e-5 * max(1.0, sc) and torch.isfinite(out).all().item()
                bad += 0 if ok else 1
                if not ok or (h == 180 and nprod == 6):
                    print(f"{'OK ' if ok else 'BAD'} shape {(n, c, h, w, cout, D)} sigma {sigma} nprod {nprod} heads {heads}: "
                          f"max|il2 - il| {err:.3e} (scale {sc:.2f})", flush=True)
print("mismatches:", bad, flush=True)


def timed(fn):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


h, w = 180, 320
w33, b = r(64, 64, 3, 3) * 0.05, r(64) * 0.1
for n in (2, 4):
    x = r(n, 64, h, w)
    xil = ops.to_il8(x)
    px = n * h * w
    for sigma in (0.5, 1.5, 4.0):
        off = r(n, 144, h, w) * sigma
        mask = torch.rand(n, 72, h, w, device=dev)
        hd = heads_of(n, 8, h, w, sigma)
        res = {}
        for impl in ("il", "il2"):
            for nprod in (6, 9):
                res[f"{impl}_{nprod}"] = timed(lambda: run(impl, xil, off, mask, w33, b, 8, nprod=nprod))
                res[f"{impl}_{nprod}_heads"] = timed(lambda: run(impl, xil, hd, None, w33, b, 8, nprod=nprod, heads=True))
        print(f"n={n} sigma={sigma}: " + "  ".join(f"{k} {v:.1f} us ({1376.0 * px / v / 1e3 / 8000:.3f})" for k, v in res.items()),
              flush=True)
sys.exit(1 if bad else 0)
