#!/usr/bin/env python3
"""How long is a DCNv2 launch whose workgroups all do exactly 1 / 2 tiles (256 / 512 tiles on 256 CUs), against the bench shape's
460 tiles (204 workgroups do 2, 52 do 1)?  Gives the per-k-step cost and what a perfectly balanced 460-tile launch would take."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eavsr_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
torch.manual_seed(0)


def t_of(n, h, w, reps=15, inner=10):
    x = torch.randn(n, 64, h, w, device=dev)
    xil = ops.to_il8(x)
    ident = torch.tensor([1.0, 0, 0, 1.0], device=dev).repeat(8).view(1, 32, 1, 1)
    hd = torch.cat([torch.randn(n, 32, h, w, device=dev) * 0.25 + ident, torch.randn(n, 16, h, w, device=dev) * 0.5,
                    torch.sigmoid(torch.randn(n, 72, h, w, device=dev))], 1)
    wt, b = torch.randn(64, 64, 3, 3, device=dev) * 0.05, torch.randn(64, device=dev) * 0.1
    f = lambda: ops.dcnv2_il(xil, hd, None, wt, b, 8, heads=True, mask_activated=True)
    for _ in range(5):
        f()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    ts.sort()
    return ts[len(ts) // 2]


for rnd in range(2):
    t1 = t_of(1, 128, 512)       # 16 x 16 tiles = 256: one tile per workgroup
    t2 = t_of(2, 128, 512)       # 512: two tiles per workgroup
    t3 = t_of(3, 128, 512)       # 768
    tb = t_of(2, 180, 320)       # the bench shape: 460 tiles
    per_tile = t2 - t1
    print(f"round {rnd}: 256 tiles {t1:.1f} us, 512 tiles {t2:.1f} us, 768 tiles {t3:.1f} us, bench shape (460 tiles) {tb:.1f} us; "
          f"one more tile per workgroup = {per_tile:.1f} us ({per_tile / 36:.2f} us per k-step); a balanced 460-tile launch "
          f"(64.7 k-steps per workgroup) would take ~{t1 + per_tile * (64.7 - 36) / 36:.1f} us, with 66 k-steps {t1 + per_tile * 30 / 36:.1f} us", flush=True)
