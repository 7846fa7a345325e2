"""GPU tests of the 16-bit (bf16 / fp16) residual backbone (BASELINE configs[2], [4]): the NHWC MFMA conv, the
layout converters and the RCAB tail against fp32 references computed on the SAME 16-bit-rounded inputs,
and a whole RCAGroup / EAVSRP forward in 16-bit backbone mode judged by relative error / PSNR against the
fp32 oracle (the tolerance for reduced precision is a PSNR, as BASELINE.json's metric says)."""
from argparse import Namespace

import pytest
import torch
import torch.nn.functional as F

from oracle import eavsr_oracle as O
from tests import helpers as H
from tests.golden import cases

pytestmark = pytest.mark.gpu
DT = {"bf16": torch.bfloat16, "fp16": torch.float16}
EPS = {"bf16": 2.0 ** -8, "fp16": 2.0 ** -11}


@pytest.fixture(scope="module")
def ops(cuda):
    from eavsr_amd import ops as _ops
    _ops.lib()
    return _ops


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(1, 8, 32), (2, 19, 37), (1, 33, 70), (1, 5, 3)])
def test_layout_converters_round_trip(ops, cuda, dt, shape):
    n, h, w = shape
    x = cases.randn(1, n, 64, h, w)
    xh = ops.to_nhwc_h16(x.to(cuda), dt)
    assert xh.shape == (n, h, w, 64) and xh.dtype == DT[dt]
    assert torch.equal(xh.cpu(), x.permute(0, 2, 3, 1).to(DT[dt]))
    res = cases.randn(2, n, 64, h, w)
    back = ops.from_nhwc_h16(xh, residual=res.to(cuda)).cpu()
    assert H.maxabs(back, x.to(DT[dt]).float() + res) <= 1e-6


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("shape,act", [((1, 8, 32), "lrelu"), ((2, 19, 37), "lrelu"), ((1, 45, 80), None), ((3, 7, 5), "relu")])
def test_conv3x3_c64_h16_act_and_pixel_shuffle_vs_fp32_on_rounded_inputs(ops, cuda, dt, shape, act):
    """the upsampling tail in the 16-bit modes (eavsrp_model.py:343-357): conv_hr's LeakyReLU form and the conv 64 -> 256 +
    PixelShuffle(2) + activation stage as four slices of the backbone kernel, against torch on the same rounded operands"""
    n, h, w = shape
    x = cases.randn(11, n, 64, h, w).to(DT[dt])
    fact = {"lrelu": lambda z: F.leaky_relu(z, 0.1), "relu": F.relu, None: lambda z: z}[act]
    xh = x.permute(0, 2, 3, 1).contiguous().to(cuda)
    wt = cases.randn(12, 64, 64, 3, 3, scale=1.0 / 24.0)
    b = cases.randn(13, 64, scale=0.1)
    ref = fact(F.conv2d(x.float(), wt.to(DT[dt]).float(), b, 1, 1))
    got = ops.conv3x3_c64_h16_act(xh, wt.to(cuda), b.to(cuda), act=act, slope=0.1).float().permute(0, 3, 1, 2).cpu()
    assert H.maxabs(got, ref) <= (EPS[dt] * 1.01) * max(1.0, ref.abs().max().item()) + 1e-5
    w4 = cases.randn(14, 256, 64, 3, 3, scale=1.0 / 24.0)
    b4 = cases.randn(15, 256, scale=0.1)
    ref = fact(F.pixel_shuffle(F.conv2d(x.float(), w4.to(DT[dt]).float(), b4, 1, 1), 2))
    out = ops.conv3x3_c64_h16_act(xh, w4.to(cuda), b4.to(cuda), act=act, slope=0.1, pixel_shuffle2=True)
    assert out.shape == (n, 2 * h, 2 * w, 64)
    got = out.float().permute(0, 3, 1, 2).cpu()
    assert H.maxabs(got, ref) <= (EPS[dt] * 1.01) * max(1.0, ref.abs().max().item()) + 1e-5


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("n,h,w,cins,cout,act", [(1, 16, 32, (64,), 64, "relu"), (2, 19, 36, (64, 64), 64, "lrelu"),
                                                 (1, 45, 80, (64, 64, 64, 64, 64), 64, "lrelu"), (1, 33, 40, (128,), 256, None),
                                                 (1, 20, 44, (256,), 64, None), (2, 7, 8, (64, 16), 96, "relu"),
                                                 (1, 16, 32, (32,), 40, "lrelu")])
def test_conv3x3_h16g_vs_fp64_on_rounded_inputs(ops, cuda, dt, n, h, w, cins, cout, act):
    """the generic 3x3 convolution of the 16-bit modes (first conv of a backbone over the concatenated feature lists, encoder
    layers): fp32 NCHW in / out, operands rounded once to the 16-bit type, fp32 accumulation -- against an fp64 convolution
    of the same rounded operands; and `ops.conv2d` routes to it only while the mode is set"""
    srcs = [cases.randn(31 + i, n, c, h, w) for i, c in enumerate(cins)]
    cin = sum(cins)
    wt = cases.randn(41, cout, cin, 3, 3, scale=1.0 / (3.0 * cin ** 0.5))
    b = cases.randn(42, cout, scale=0.1)
    fact = {"lrelu": lambda z: F.leaky_relu(z, 0.1), "relu": F.relu, None: lambda z: z}[act]
    ref = fact(F.conv2d(torch.cat(srcs, 1).to(DT[dt]).double(), wt.to(DT[dt]).double(), b.double(), 1, 1))
    assert ops.CONV3_H16 is None
    ops.set_conv3_h16(dt)
    try:
        with torch.no_grad(), ops.profile() as prof:
            got = ops.conv2d([s_.to(cuda) for s_ in srcs], wt.to(cuda), b.to(cuda), act=act, slope=0.1).cpu()
    finally:
        ops.set_conv3_h16(None)
    assert f"conv3x3_{cin}to{cout}_h16g" in prof.summary()
    assert got.shape == (n, cout, h, w) and got.dtype == torch.float32
    assert H.maxabs(got.double(), ref) <= 3e-5 * max(1.0, ref.abs().max().item())      # exact 16-bit products, fp32 accumulation


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("n,cin,cout,h,w,act", [(2, 8, 32, 24, 40, "relu"), (1, 32, 64, 45, 80, "relu"), (1, 64, 32, 17, 33, "relu"),
                                               (3, 32, 16, 9, 12, None), (12, 32, 64, 48, 80, "relu")])
def test_conv7x7_h16x1_vs_fp64_on_rounded_inputs(ops, cuda, dt, n, cin, cout, h, w, act):
    """SPyNet's 7x7 layers in the 16-bit modes (one operand plane of the bf16x6 kernel): against an fp64 convolution of the
    operands rounded to the 16-bit type; the 16 -> 2 flow head is not routed (it stays exact)"""
    x = cases.randn(51, n, cin, h, w)
    wt = cases.randn(52, cout, cin, 7, 7, scale=1.0 / (7.0 * cin ** 0.5))
    b = cases.randn(53, cout, scale=0.1)
    ref = F.conv2d(x.to(DT[dt]).double(), wt.to(DT[dt]).double(), b.double(), 1, 3)
    ref = F.relu(ref) if act == "relu" else ref
    ops.set_conv3_h16(dt)
    try:
        with torch.no_grad(), ops.profile() as prof:
            got = ops.conv2d(x.to(cuda), wt.to(cuda), b.to(cuda), act=act).cpu()
            w2 = cases.randn(54, 2, 16, 7, 7, scale=0.05)
            ops.conv2d(cases.randn(55, 1, 16, h, w).to(cuda), w2.to(cuda), None)
    finally:
        ops.set_conv3_h16(None)
    names = prof.summary()
    assert f"conv7x7_{cin}to{cout}_h16x1" in names and "conv7x7_16to2_x6" in names
    assert H.maxabs(got.double(), ref) <= 3e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("shape,res", [((1, 8, 32), True), ((2, 19, 37), False), ((1, 45, 80), True), ((3, 7, 5), True)])
def test_conv3x3_c64to3_h16_vs_fp64_on_rounded_inputs(ops, cuda, dt, shape, res):
    """conv_last of the tail (eavsrp_model.py:359-360) in the 16-bit modes: fp32 NCHW out (+ the skip image)"""
    n, h, w = shape
    x = cases.randn(21, n, 64, h, w).to(DT[dt])
    wt = cases.randn(22, 3, 64, 3, 3, scale=1.0 / 24.0)
    b = cases.randn(23, 3, scale=0.1)
    r = cases.randn(24, n, 3, h, w) if res else None
    ref = F.conv2d(x.double(), wt.to(DT[dt]).double(), b.double(), 1, 1) + (r.double() if res else 0.0)
    got = ops.conv3x3_c64to3_h16(x.permute(0, 2, 3, 1).contiguous().to(cuda), wt.to(cuda), b.to(cuda),
                                 residual=r.to(cuda) if res else None).cpu()
    assert got.shape == (n, 3, h, w) and got.dtype == torch.float32
    assert H.maxabs(got.double(), ref) <= 2e-5 * max(1.0, ref.abs().max().item())      # fp32 accumulation of exact 16-bit products


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("shape,relu,part", [((1, 8, 32), False, False), ((2, 19, 37), True, True),
                                             ((1, 45, 80), True, False), ((1, 64, 64), False, True),
                                             ((3, 7, 5), True, True), ((1, 264, 512), True, True), ((2, 272, 512), False, True)])
def test_conv3x3_c64_h16_vs_fp32_on_rounded_inputs(ops, cuda, dt, shape, relu, part):
    n, h, w = shape
    x = cases.randn(1, n, 64, h, w).to(DT[dt])
    wt = cases.randn(2, 64, 64, 3, 3, scale=1.0 / 24.0)
    b = cases.randn(3, 64, scale=0.1)
    ref = F.conv2d(x.float(), wt.to(DT[dt]).float(), b, 1, 1)
    ref = F.relu(ref) if relu else ref
    xh = x.permute(0, 2, 3, 1).contiguous().to(cuda)
    out = ops.conv3x3_c64_h16(xh, wt.to(cuda), b.to(cuda), relu=relu, chan_partial=part)
    if part:
        out, p = out
        assert H.maxabs(p.sum(1).cpu(), ref.sum(dim=(2, 3))) <= 2e-3 * max(1.0, ref.sum(dim=(2, 3)).abs().max().item())
    got = out.float().permute(0, 3, 1, 2).cpu()
    # one rounding of the fp32 result to 16 bits (+ fp32 summation-order noise)
    assert H.maxabs(got, ref) <= (EPS[dt] * 1.01) * max(1.0, ref.abs().max().item()) + 1e-5


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("shape,couts", [((1, 16, 32), (32, 16, 72)), ((2, 19, 37), (32, 16, 72)), ((1, 45, 80), (120,)),
                                         ((1, 5, 3), (4, 2, 9)), ((3, 33, 70), (128,))])
def test_conv5x5_c64_h16_heads_vs_fp32_on_rounded_operands(ops, cuda, dt, shape, couts):
    """the predictor's 5x5 heads in the 16-bit modes (networks.py:283-285 as one convolution): 16-bit NHWC input, weights
    rounded once, fp32 accumulation and fp32 NCHW output -- against an fp64 convolution of the SAME rounded operands (only the
    summation order differs: tolerance a few fp32 roundings of the 1600-term sum)"""
    n, h, w = shape
    x = cases.randn(1, n, 64, h, w).to(DT[dt])
    ws = [cases.randn(2 + i, c, 64, 5, 5, scale=1.0 / 40.0) for i, c in enumerate(couts)]
    bs = [cases.randn(9 + i, c, scale=0.1) for i, c in enumerate(couts)]
    ref = F.conv2d(x.double(), torch.cat(ws, 0).to(DT[dt]).double(), torch.cat(bs, 0).double(), 1, 2).float()
    xh = x.permute(0, 2, 3, 1).contiguous().to(cuda)
    out = ops.conv5x5_c64_h16(xh, [w_.to(cuda) for w_ in ws], [b_.to(cuda) for b_ in bs])
    assert out.dtype == torch.float32 and tuple(out.shape) == (n, sum(couts), h, w)
    assert H.maxabs(out.cpu(), ref) <= 2e-5 * max(1.0, ref.abs().max().item())


def test_conv5x5_c64_h16_bad_arguments(ops, cuda):
    z = lambda *s_, dt=torch.bfloat16: torch.zeros(*s_, device=cuda, dtype=dt)
    with pytest.raises(NotImplementedError):
        ops.conv5x5_c64_h16(z(1, 8, 8, 32), z(16, 32, 5, 5, dt=torch.float32), None)            # 64 input channels only
    with pytest.raises(NotImplementedError):
        ops.conv5x5_c64_h16(z(1, 8, 8, 64), z(130, 64, 5, 5, dt=torch.float32), None)           # cout <= 128
    with pytest.raises(RuntimeError):
        ops.conv5x5_c64_h16(z(1, 8, 8, 64, dt=torch.float32), z(16, 64, 5, 5, dt=torch.float32), None)   # fp32 input


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_scale_residual_h16(ops, cuda, dt):
    r, x = cases.randn(1, 2, 64, 9, 13).to(DT[dt]), cases.randn(2, 2, 64, 9, 13).to(DT[dt])
    s = cases.rand(3, 2, 64)
    ref = (r.float() * s.view(2, 64, 1, 1) + x.float()).to(DT[dt]).float()
    out = ops.scale_residual_h16(r.permute(0, 2, 3, 1).contiguous().to(cuda), s.to(cuda),
                                 x.permute(0, 2, 3, 1).contiguous().to(cuda))
    assert H.maxabs(out.float().permute(0, 3, 1, 2).cpu(), ref) <= EPS[dt] * 4


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_rcagroup_16bit_backbone_vs_fp32_oracle(ops, cuda, dt):
    from eavsr_amd import networks as Nw
    sd = H.filled(H.rcagroup_shapes("g.", 4), "trained_like")
    x = cases.randn(5, 2, 64, 36, 44)
    ref = O.rca_group(sd, "g.", x, 4)
    grp = Nw.RCAGroup(64, 64, nb=4)
    grp.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=True)
    grp = grp.to(cuda).eval()
    try:
        Nw.set_backbone_dtype(dt)
        with torch.no_grad():
            out = grp(x.to(cuda)).cpu()
    finally:
        Nw.set_backbone_dtype(None)
    rel = H.maxabs(out, ref) / ref.abs().max().item()
    assert rel <= (4e-2 if dt == "bf16" else 6e-3), rel
    # the default computes each RCAB's attention before its second convolution (ops.ca_scale_pre_h16); the four-launch form agrees
    try:
        Nw.set_backbone_dtype(dt)
        Nw.set_rcab_h16_pre(False)
        with torch.no_grad(), ops.profile() as prof:
            out0 = grp(x.to(cuda)).cpu()
        assert "scale_residual_h16" in set(prof.summary())
    finally:
        Nw.set_rcab_h16_pre(True)
        Nw.set_backbone_dtype(None)
    assert H.maxabs(out0, out) <= 8 * EPS[dt] * ref.abs().max().item()
    assert H.maxabs(out0, ref) / ref.abs().max().item() <= (4e-2 if dt == "bf16" else 6e-3)
    # the same group with each RCAB's two convolutions as ONE launch (csrc/rcab_h16.hip, opt-in): same r, channel sums by another
    # summation tree -- the group output may differ by a 16-bit rounding flip here and there, never by more
    if ops.lab_available():      # (a retired schedule: lab build only)
        try:
            Nw.set_backbone_dtype(dt)
            Nw.set_rcab_h16_fused(True)
            with torch.no_grad(), ops.profile() as prof:
                out1 = grp(x.to(cuda)).cpu()
            assert "rcab_convs_h16" in set(prof.summary())
        finally:
            Nw.set_rcab_h16_fused(False)
            Nw.set_backbone_dtype(None)
        assert H.maxabs(out1, out) <= 4 * EPS[dt] * ref.abs().max().item()
        assert H.maxabs(out1, ref) / ref.abs().max().item() <= (4e-2 if dt == "bf16" else 6e-3)
    with torch.no_grad():
        exact = grp(x.to(cuda)).cpu()          # back to the exact fp32 path
    assert H.maxabs(exact, ref) <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_end_to_end_16bit_backbone_psnr(ops, cuda, dt):
    from eavsr_amd import networks as Nw
    from eavsr_amd.eavsrp_model import EAVSRP
    gold = H.golden("g8_e2e_x2_trained_like")
    net = EAVSRP(Namespace(predict=False, n_frame=7, n_flow=5, scale=2), None)
    net.load_state_dict(H.filled(H.model_shapes("x2"), "trained_like"), strict=True)
    net = net.to(cuda).eval()
    clip = cases.g8_clip().to(cuda)
    try:
        Nw.set_backbone_dtype(dt)
        with torch.no_grad():
            y = net(clip).cpu()
    finally:
        Nw.set_backbone_dtype(None)
    sub = cases.subsample(y)
    mse = ((sub.clamp(0, 1) * 255).round() - (gold["sub"].clamp(0, 1) * 255).round()).div(255).pow(2).mean().item()
    psnr = float("inf") if mse == 0 else -10 * torch.log10(torch.tensor(mse)).item()
    assert psnr >= (50.0 if dt == "bf16" else 60.0), psnr


# ---- 16-bit alignment kernels (round 2): IL8 16-bit warp output + DCNv2 on the 16-bit MFMA --------------------------------
def _dcn_inputs16(n, c, h, w, cout, dg, sigma, seed=0):
    x = cases.randn(seed + 1, n, c, h, w)
    off = cases.randn(seed + 2, n, dg * 18, h, w, scale=sigma)
    mask = cases.rand(seed + 3, n, dg * 9, h, w)
    wt = cases.randn(seed + 4, cout, c, 3, 3, scale=1.0 / (c * 9) ** 0.5)
    b = cases.randn(seed + 5, cout, scale=0.1)
    return x, off, mask, wt, b


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_to_il8_h16_layout_and_rounding(ops, cuda, dt):
    x = cases.randn(3, 2, 24, 7, 9)
    il = ops.to_il8_h16(x.to(cuda), dt).cpu()
    assert il.shape == (2, 3, 7, 9, 8) and il.dtype == DT[dt]
    assert torch.equal(il, x.view(2, 3, 8, 7, 9).permute(0, 1, 3, 4, 2).to(DT[dt]))


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("sigma", [0.5, 2.0, 8.0])
@pytest.mark.parametrize("shape", [(1, 64, 24, 40, 64, 8), (2, 64, 13, 37, 64, 8), (1, 16, 9, 33, 32, 2), (2, 64, 45, 80, 64, 8)])
def test_dcnv2_il16_vs_oracle_on_rounded_operands(ops, cuda, shape, sigma, dt):
    """the 16-bit DCNv2 against the fp32 oracle fed the SAME 16-bit-rounded features and weights: what remains is one
    rounding of each blended sample to 16 bits (random signs over 9 c terms) and fp32 summation order"""
    n, c, h, w, cout, dg = shape
    x, off, mask, wt, b = _dcn_inputs16(n, c, h, w, cout, dg, sigma)
    xr, wr = x.to(DT[dt]).float(), wt.to(DT[dt]).float()
    ref = O.dcnv2(xr, off, mask, wr, b, 1, 1, 1, 1, dg)
    out = ops.dcnv2_il16(ops.to_il8_h16(x.to(cuda), dt), off.to(cuda), mask.to(cuda), wt.to(cuda), b.to(cuda), dg).cpu()
    rel = H.maxabs(out, ref) / max(1.0, ref.abs().max().item())
    assert rel <= (8e-3 if dt == "bf16" else 1e-3), rel


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_dcnv2_il16_heads_mode(ops, cuda, dt):
    n, h, w, D = 2, 21, 44, 8
    c = 8 * D
    x = cases.randn(11, n, c, h, w)
    heads = torch.cat([cases.randn(12, n, 4 * D, h, w, scale=0.4) + torch.tensor([1.0, 0, 0, 1.0]).repeat(D).view(1, 4 * D, 1, 1),
                       cases.randn(13, n, 2 * D, h, w, scale=1.5), cases.randn(14, n, 9 * D, h, w, scale=2.0)], 1)
    wt = cases.randn(15, 64, c, 3, 3, scale=1.0 / (c * 9) ** 0.5)
    b = cases.randn(16, 64, scale=0.1)
    off = O.affine_offsets(heads[:, :4 * D], heads[:, 4 * D:6 * D], D)
    mask = torch.sigmoid(heads[:, 6 * D:])
    ref = O.dcnv2(x.to(DT[dt]).float(), off, mask, wt.to(DT[dt]).float(), b, 1, 1, 1, 1, D)
    xil = ops.to_il8_h16(x.to(cuda), dt)
    out = ops.dcnv2_il16(xil, heads.to(cuda), None, wt.to(cuda), b.to(cuda), D, heads=True).cpu()
    rel = H.maxabs(out, ref) / max(1.0, ref.abs().max().item())
    assert rel <= (8e-3 if dt == "bf16" else 1e-3), rel
    out2 = ops.dcnv2_il16(xil, off.to(cuda), mask.to(cuda), wt.to(cuda), b.to(cuda), D).cpu()
    assert H.maxabs(out, out2) <= (4e-3 if dt == "bf16" else 5e-4) * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_flow_warp_pair_16bit_il8_output(ops, cuda, dt):
    xa, xb = cases.randn(1, 2, 64, 13, 37), cases.randn(2, 2, 64, 13, 37)
    flow = cases.randn(3, 2, 2, 13, 37, scale=2.0)
    a32, b32 = ops.flow_warp_pair(xa.to(cuda), xb.to(cuda), flow.to(cuda))
    a16, b16 = ops.flow_warp_pair(xa.to(cuda), xb.to(cuda), flow.to(cuda), b_il8=dt)
    assert torch.equal(a32, a16) and b16.dtype == DT[dt] and b16.shape == (2, 8, 13, 37, 8)
    want = ops.to_il8_h16(b32, dt).float().cpu()
    # the octet loop may contract multiply-adds differently (one fp32 rounding) before the 16-bit rounding: <= 1 ulp
    assert H.maxabs(b16.float().cpu(), want) <= EPS[dt] * 2 * max(1.0, xb.abs().max().item())


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_multiadstn_16bit_alignment_vs_reference_golden(ops, cuda, dt):
    """MultiAdSTN with the 16-bit warp + DCNv2 (the backbone-dtype switch turns them on) against the reference's fp32 output
    (golden G5), by relative error -- the tolerance of the 16-bit configs is a PSNR, not 1e-3"""
    from eavsr_amd import networks as Nw
    gold = H.golden("g5_multiadstn_trained_like")
    sd = H.filled(H.multiadstn_shapes("g5.align."), "trained_like")
    m = Nw.MultiAdSTN(Namespace(predict=False, n_frame=7, n_flow=5, scale=4), 64, 64, deformable_groups=8)
    m.load_state_dict({k[len("g5.align."):]: v for k, v in sd.items()}, strict=True)
    m = m.to(cuda).eval()
    nbr, ref, fp, flow = cases.g5_inputs()
    try:
        Nw.set_backbone_dtype(dt)
        with torch.no_grad(), ops.profile() as prof:
            out = m([t.to(cuda) for t in nbr], [t.to(cuda) for t in ref], fp.to(cuda), flow.to(cuda)).cpu()
        names = set(prof.summary())
    finally:
        Nw.set_backbone_dtype(None)
    assert "dcnv2_il16_heads" in names and "dcnv2_il_heads" not in names
    assert "conv5x5_64to120_h16" in names and "conv5x5_64to120_wino" not in names, names     # the heads ran in 16 bits too
    rel = H.maxabs(out, gold["out"]) / gold["out"].abs().max().item()
    assert rel <= (2e-2 if dt == "bf16" else 3e-3), rel


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(1, 8, 32), (2, 19, 37), (1, 45, 80), (3, 7, 5), (2, 1, 40), (1, 33, 1), (1, 1, 1), (2, 256, 256), (1, 540, 960)])
def test_rcab_attention_before_the_second_convolution(ops, cuda, dt, shape):
    """eavsr_ca_scale_pre_h16 + eavsr_conv3x3_c64_h16_res (round 5): the attention of an RCAB (CALayer, networks.py:444-447) from
    border-corrected channel sums of the second convolution's INPUT -- sum_o conv(t)[co][o] = sum W (T - R(ky) - C(kx) + X) -- must
    equal the attention computed from the convolution's OUTPUT (ca_scale on its channel sums), and `x + scale * conv(t)` in the
    convolution's epilogue must equal scale_residual_h16 of the separately stored r up to one 16-bit rounding (r is no longer
    rounded before the product).  Single rows / columns / pixels (where the excluded border lines coincide), ragged tiles,
    several samples, the per-workgroup row layout of the sums (540 x 960)."""
    n, h, w = shape
    x = cases.randn(41, n, 64, h, w).to(DT[dt])
    xh = x.permute(0, 2, 3, 1).contiguous().to(cuda)
    w1, w2 = cases.randn(42, 64, 64, 3, 3, scale=1.0 / 24.0).to(cuda), cases.randn(43, 64, 64, 3, 3, scale=1.0 / 24.0).to(cuda)
    b1, b2 = cases.randn(44, 64, scale=0.1).to(cuda), cases.randn(45, 64, scale=0.1).to(cuda)
    a_w, a_b = cases.randn(46, 4, 64, scale=0.5).to(cuda), cases.randn(47, 4, scale=0.1).to(cuda)
    c_w, c_b = cases.randn(48, 64, 4, scale=0.5).to(cuda), cases.randn(49, 64, scale=0.1).to(cuda)
    # the four-launch form
    t = ops.conv3x3_c64_h16(xh, w1, b1, relu=True)
    r, rpart = ops.conv3x3_c64_h16(t, w2, b2, chan_partial=True)
    scale_post = ops.ca_scale(rpart, h * w, a_w, a_b, c_w, c_b)
    y_post = ops.scale_residual_h16(r, scale_post, xh)
    # attention first, tail in the second convolution
    t2, tpart = ops.conv3x3_c64_h16(xh, w1, b1, relu=True, chan_partial=True)
    assert torch.equal(t2, t)
    scale_pre = ops.ca_scale_pre_h16(t2, tpart, w2, b2, a_w, a_b, c_w, c_b)
    y_pre = ops.conv3x3_c64_h16_res(t2, w2, b2, xh, scale_pre)
    # round 6: the border-line sums as a by-product of the first convolution's epilogue (eavsr_conv3x3_c64_h16_b) instead of a launch
    # of their own: same t and plane sums bit for bit, the pieces equal the border lines of t, the attention equal to summation order
    t3, tpart3, pieces = ops.conv3x3_c64_h16(xh, w1, b1, relu=True, chan_partial=True, border=True)
    assert torch.equal(t3, t) and torch.equal(tpart3, tpart)
    tf, pd = t.float().cpu(), pieces.data.cpu()
    lines = [tf[:, 0].sum(1), tf[:, h - 1].sum(1), tf[:, :, 0].sum(1), tf[:, :, w - 1].sum(1)]      # (n, 64) each
    for bi, (want, cnt) in enumerate(zip(lines, (pieces.p_rows, pieces.p_rows, pieces.p_cols, pieces.p_cols))):
        got = pd[:, bi, :cnt].sum(1)
        assert H.maxabs(got, want) <= 2e-5 * max(1.0, want.abs().max().item()), (bi, H.maxabs(got, want))
    with ops.profile() as prof:
        scale_pc = ops.ca_scale_pre_h16(t2, tpart, w2, b2, a_w, a_b, c_w, c_b, border=pieces)
    assert prof.summary()["ca_scale_pre_h16"]["calls"] == 1
    assert H.maxabs(scale_pc.cpu(), scale_pre.cpu()) <= 2e-6
    # the means behind the two attentions: fp32 sums of the same products in another order (+ r's rounding in the old form)
    assert H.maxabs(scale_pre.cpu(), scale_post.cpu()) <= (3e-3 if dt == "bf16" else 5e-4)
    ref = xh.float() + r.float() * scale_post[:, None, None, :]      # what the old form rounds
    tol = EPS[dt] * 2.0 * max(1.0, ref.abs().max().item())
    assert H.maxabs(y_pre.float().cpu(), ref.cpu()) <= tol
    assert H.maxabs(y_pre.float().cpu(), y_post.float().cpu()) <= tol


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(1, 8, 32), (2, 19, 37), (1, 45, 80), (3, 7, 5), (1, 64, 96), (2, 256, 256)])
def test_rcab_convs_in_one_launch_equal_the_two_launches(ops, cuda, dt, shape):
    """eavsr_rcab_convs_h16 (csrc/rcab_h16.hip): conv3x3 -> ReLU -> conv3x3 of RCABlock.forward (networks.py:461-462) as ONE
    launch -- streamed weights, the intermediate rounded to 16 bits in LDS and zero outside the image.  r must equal the two
    eavsr_conv3x3_c64_h16 launches BIT FOR BIT (same operands, same k-step order), on single tiles, ragged edges (the intermediate's
    zero padding), several samples and a multi-tile-per-workgroup launch; the channel sums (another summation tree) to rounding;
    and against torch on the same rounded operands."""
    n, h, w = shape
    x = cases.randn(31, n, 64, h, w).to(DT[dt])
    xh = x.permute(0, 2, 3, 1).contiguous().to(cuda)
    w1, w2 = cases.randn(32, 64, 64, 3, 3, scale=1.0 / 24.0), cases.randn(33, 64, 64, 3, 3, scale=1.0 / 24.0)
    b1, b2 = cases.randn(34, 64, scale=0.1), cases.randn(35, 64, scale=0.1)
    gw1, gw2, gb1, gb2 = w1.to(cuda), w2.to(cuda), b1.to(cuda), b2.to(cuda)
    t = ops.conv3x3_c64_h16(xh, gw1, gb1, relu=True)
    r_two, part_two = ops.conv3x3_c64_h16(t, gw2, gb2, chan_partial=True)
    r_one, part_one = ops.rcab_convs_h16(xh, gw1, gb1, gw2, gb2, chan_partial=True)
    assert r_one.shape == r_two.shape and r_one.dtype == r_two.dtype
    assert torch.equal(r_one, r_two), H.maxabs(r_one.float().cpu(), r_two.float().cpu())
    s_one, s_two = part_one.sum(1).cpu(), part_two.sum(1).cpu()
    want = r_two.float().sum(dim=(1, 2)).cpu()
    scale = max(1.0, want.abs().max().item())
    assert H.maxabs(s_one, want) <= 2e-5 * scale * (h * w) ** 0.5 and H.maxabs(s_one, s_two) <= 2e-5 * scale * (h * w) ** 0.5
    r_nop = ops.rcab_convs_h16(xh, gw1, None, gw2, None, chan_partial=False)      # no biases, no sums
    t0 = ops.conv3x3_c64_h16(xh, gw1, None, relu=True)
    assert torch.equal(r_nop, ops.conv3x3_c64_h16(t0, gw2, None))
    if h * w <= 64 * 96:
        tt = F.relu(F.conv2d(x.float(), w1.to(DT[dt]).float(), b1, 1, 1)).to(DT[dt]).float()
        ref = F.conv2d(tt, w2.to(DT[dt]).float(), b2, 1, 1)
        got = r_one.float().permute(0, 3, 1, 2).cpu()
        assert H.maxabs(got, ref) <= (EPS[dt] * 1.5) * max(1.0, ref.abs().max().item()) + 1e-4
