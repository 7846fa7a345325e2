#!/usr/bin/env python3
"""Random-shape cross-check of the Winograd kernels (MODE=winograd: F(2x2,3x3); MODE=winograd4, the default: F(4x4,3x3)
and, with K=5, F(2x2,5x5)) against the direct kernel (both on the GPU): ragged heights,
widths that are multiples of 4 but not of 32, odd batch sizes, 1-5 sources, output channels that do not fill a
64-channel tile, with / without bias, activation, residual and channel sums."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from eavsr_amd import ops
dev = torch.device("cuda:0")
rng = random.Random(int(os.environ.get("SEED", 0)))
MODE = os.environ.get("MODE", "winograd4")
K = int(os.environ.get("K", 3))
TOL = 2e-5 if MODE == "winograd" else 8e-5
worst = 0.0
n_cases = int(os.environ.get("CASES", 40))
for case in range(n_cases):
    h = rng.choice([8, 9, 17, 33, 64, 90, 100, 133, 180, 200])
    w = 4 * rng.randint(2, 90)
    tiles = ((h + 7) // 8) * ((w + 31) // 32)
    n = max(1, (200 + tiles - 1) // tiles + rng.randint(0, 2))
    while n * h * w * 64 * 4 > 400e6:   # keep tensors modest
        n = max(1, n // 2); 
        if n * tiles < 192: break
    if n * tiles < 192:
        continue
    nsrc = rng.choice([1, 1, 2, 3, 5])
    chans = [(8 if K == 3 else 4) * rng.randint(1, 8 if nsrc > 1 else 16) for _ in range(nsrc)]
    cout = rng.choice([8, 24, 40, 64, 64, 72, 120, 128, 256])
    act = rng.choice([None, "relu", "lrelu"])
    use_b, use_res, use_part = rng.random() < 0.8, rng.random() < 0.4 and act is None, rng.random() < 0.4
    g = torch.Generator(device="cpu").manual_seed(case)
    srcs = [torch.randn(n, c, h, w, generator=g).to(dev) for c in chans]
    cin = sum(chans)
    wt = (torch.randn(cout, cin, K, K, generator=g) / (cin * K * K) ** 0.5).to(dev)
    b = torch.randn(cout, generator=g).to(dev) if use_b else None
    res = torch.randn(n, cout, h, w, generator=g).to(dev) if use_res else None
    outs = {}
    for mode in ("direct", MODE):
        ops.set_conv_mode(mode)
        with ops.profile() as prof:
            o = ops.conv2d(srcs, wt, b, act=act, slope=0.1, residual=res, chan_partial=use_part)
        names = list(prof.summary())
        outs[mode] = (o, names)
    (od, nd), (ow, nw) = outs["direct"], outs[MODE]
    if "_wino" not in nw[0]:
        continue   # below this mode's tile threshold: the direct kernel ran
    if use_part:
        (od, pd), (ow, pw) = od, ow
        sd, sw = pd.sum(1), pw.sum(1)
        assert (sd - sw).abs().max().item() <= 2e-5 * sd.abs().max().item() + 5e-3, "channel sums"
    err = (od - ow).abs().max().item() / max(1.0, od.abs().max().item())
    worst = max(worst, err)
    print(f"case {case:3d}: n {n:3d} {h:3d}x{w:3d} cin {chans} cout {cout:3d} act {act} bias {use_b} res {use_res} part {use_part}: rel diff {err:.2e}", flush=True)
    assert err <= TOL, err
print("worst relative difference", worst)
