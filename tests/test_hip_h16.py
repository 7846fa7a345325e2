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
@pytest.mark.parametrize("shape,relu,part", [((1, 8, 32), False, False), ((2, 19, 37), True, True),
                                             ((1, 45, 80), True, False), ((1, 64, 64), False, True),
                                             ((3, 7, 5), True, True)])
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
