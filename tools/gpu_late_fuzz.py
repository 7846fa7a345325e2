#!/usr/bin/env python3
"""Random-shape cross-checks (all on the GPU, against other kernels / torch on the GPU) of the kernels touched late in round 2:
conv3x3 small-cout (both thread-group variants), adapt_frontend (register-tiled vs scalar kernel), the conv kernel's pixel-shuffle
store pattern, the 16-bit backbone conv and 5x5 heads (vs torch conv2d on the rounded operands), the one-launch RCAB tail,
DCNv2 IL8 in both schedules."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from eavsr_amd import ops
dev = torch.device("cuda:0")
rng = random.Random(int(os.environ.get("SEED", 0)))
CASES = int(os.environ.get("CASES", 30))
torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False
worst = {}


def note(name, err, tol):
    worst[name] = max(worst.get(name, 0.0), err)
    assert err <= tol, (name, err, tol)


def rnd(*s, g, scale=1.0):
    return (torch.randn(*s, generator=g) * scale).to(dev)


for case in range(CASES):
    g = torch.Generator(device="cpu").manual_seed(1000 + case)
    # ---- conv3x3 small cout vs torch (fp32 cudnn/miopen conv on the GPU)
    n, h, w = rng.choice([1, 2, 3, 5]), rng.choice([5, 8, 9, 17, 45, 64, 90, 180]), rng.choice([3, 7, 12, 33, 64, 80, 130, 320])
    cin, cout = rng.choice([3, 8, 16, 18, 20, 64]), rng.choice([2, 3, 4, 6])
    x, wt, b = rnd(n, cin, h, w, g=g), rnd(cout, cin, 3, 3, g=g, scale=(cin * 9) ** -0.5), rnd(cout, g=g, scale=0.1)
    res = rnd(n, cout, h, w, g=g) if rng.random() < 0.3 else None
    act = rng.choice([None, "relu", "lrelu"])
    ref = F.conv2d(x.double(), wt.double(), b.double(), 1, 1)
    ref = F.relu(ref) if act == "relu" else (F.leaky_relu(ref, 0.1) if act == "lrelu" else ref)
    ref = (ref + (res.double() if res is not None else 0)).float()
    out = ops.conv2d(x, wt, b, act=act, slope=0.1, residual=res)
    note("smallco", (out - ref).abs().max().item() / max(1.0, ref.abs().max().item()), 2e-5)
    # ---- adapt_frontend: aligned (register-tiled) vs the same data at an unaligned width (scalar kernel) vs torch
    c = rng.choice([2, 8, 64])
    h2, w2 = rng.choice([4, 9, 16, 45, 90]), 4 * rng.randint(1, 40)
    xa, ha = rnd(n, c, h2, w2, g=g), rnd(n, c, h2, w2, g=g)
    w1, b1, w2_, b2 = rnd(2 * c, 1, 3, 3, g=g, scale=0.3), rnd(2 * c, g=g, scale=0.1), rnd(c, 2, 3, 3, g=g, scale=0.3), rnd(c, g=g, scale=0.1)
    cat = torch.cat([xa, ha], 1).double()
    t1 = F.leaky_relu(F.conv2d(cat, w1.double(), b1.double(), 1, 1, groups=2 * c), 0.2)
    reff = F.leaky_relu(F.conv2d(t1, w2_.double(), b2.double(), 1, 1, groups=c), 0.2).float()
    outf = ops.adapt_frontend(xa, ha, w1, b1, w2_, b2)
    note("adapt_frontend", (outf - reff).abs().max().item() / max(1.0, reff.abs().max().item()), 2e-5)
    # ---- pixel-shuffle epilogue
    n3, h3, w3 = rng.choice([2, 3, 6]), rng.choice([64, 90, 100, 133, 180]), 4 * rng.randint(16, 80)
    co3 = rng.choice([8, 64, 256])
    x3, wt3, b3 = rnd(n3, 64, h3, w3, g=g), rnd(co3, 64, 3, 3, g=g, scale=1 / 24.0), rnd(co3, g=g, scale=0.1)
    o_ps = ops.conv2d(x3, wt3, b3, act="lrelu", slope=0.1, pixel_shuffle2=True)
    o_pl = F.pixel_shuffle(ops.conv2d(x3, wt3, b3, act="lrelu", slope=0.1), 2)
    assert torch.equal(o_ps, o_pl), "pixel shuffle epilogue"
    # ---- 16-bit convs vs torch on the rounded operands (fp64 accumulate)
    for dt in (torch.bfloat16, torch.float16):
        n4, h4, w4 = rng.choice([1, 2, 5]), rng.choice([3, 8, 9, 31, 45, 64]), rng.choice([5, 32, 33, 70, 96])
        xh = rnd(n4, 64, h4, w4, g=g).to(dt)
        wt4, b4 = rnd(64, 64, 3, 3, g=g, scale=1 / 24.0), rnd(64, g=g, scale=0.1)
        relu, part = rng.random() < 0.5, rng.random() < 0.5
        ref4 = F.conv2d(xh.double(), wt4.to(dt).double(), b4.double(), 1, 1)
        ref4 = F.relu(ref4) if relu else ref4
        o4 = ops.conv3x3_c64_h16(xh.permute(0, 2, 3, 1).contiguous(), wt4, b4, relu=relu, chan_partial=part)
        if part:
            o4, p4 = o4
            s4 = ref4.to(dt).double().sum(dim=(2, 3))
            note("h16 channel sums", (p4.sum(1).double() - s4).abs().max().item() / max(1.0, s4.abs().max().item()), 4e-3)
        eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
        note(f"conv3x3_h16 {dt}", (o4.permute(0, 3, 1, 2).double() - ref4).abs().max().item() / max(1.0, ref4.abs().max().item()), 1.01 * eps + 1e-5)
        co5 = rng.choice([6, 40, 120, 128])
        wt5, b5 = rnd(co5, 64, 5, 5, g=g, scale=1 / 40.0), rnd(co5, g=g, scale=0.1)
        ref5 = F.conv2d(xh.double(), wt5.to(dt).double(), b5.double(), 1, 2).float()
        o5 = ops.conv5x5_c64_h16(xh.permute(0, 2, 3, 1).contiguous(), wt5, b5)
        note(f"conv5x5_h16 {dt}", (o5 - ref5).abs().max().item() / max(1.0, ref5.abs().max().item()), 2e-5)
    # ---- one-launch RCAB tail vs the two launches
    n5, h5, w5 = rng.choice([1, 2, 4]), rng.choice([4, 9, 45, 180]), 4 * rng.randint(1, 80)
    c5 = rng.choice([16, 64, 128])
    r5, x5 = rnd(n5, c5, h5, w5, g=g), rnd(n5, c5, h5, w5, g=g)
    tiles = rng.randint(1, 300)
    part5 = rnd(n5, tiles, c5, g=g)
    cr = max(1, c5 // 16)
    a1, a2, a3, a4 = rnd(cr, c5, 1, 1, g=g, scale=0.2), rnd(cr, g=g, scale=0.1), rnd(c5, cr, 1, 1, g=g, scale=0.5), rnd(c5, g=g, scale=0.1)
    assert torch.equal(ops.ca_tail(r5, part5, a1, a2, a3, a4, x5),
                       ops.scale_residual(r5, ops.ca_scale(part5, h5 * w5, a1, a2, a3, a4), x5)), "ca_tail"
    # ---- DCNv2 IL8: the two schedules agree to rounding
    n6, h6, w6, D = rng.choice([1, 2]), rng.choice([7, 13, 24, 45]), rng.choice([5, 33, 40, 80]), rng.choice([1, 2, 8])
    c6 = 8 * D
    x6 = rnd(n6, c6, h6, w6, g=g)
    off6, m6 = rnd(n6, 18 * D, h6, w6, g=g, scale=rng.choice([0.5, 2.0, 8.0])), torch.rand(n6, 9 * D, h6, w6, generator=g).to(dev)
    w6_, b6 = rnd(64, c6, 3, 3, g=g, scale=(c6 * 9) ** -0.5), rnd(64, g=g, scale=0.1)
    xil = ops.to_il8(x6)
    with ops.modes(dcn_il_impl="il"):
        oa = ops.dcnv2_il(xil, off6, m6, w6_, b6, D, nprod=9)
    with ops.modes(dcn_il_impl="ws"):
        ob = ops.dcnv2_il(xil, off6, m6, w6_, b6, D, nprod=9)
    note("dcnv2 il vs ws", (oa - ob).abs().max().item() / max(1.0, oa.abs().max().item()), 2e-6)
    print(f"case {case:3d} ok", flush=True)
print("worst:", {k: f"{v:.2e}" for k, v in worst.items()})
