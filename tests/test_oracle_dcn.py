"""DCNv2 is third-party arithmetic (mmcv 1.x, absent from /root/reference): parity is UNPINNED
against the mmcv binary.  These CPU tests anchor the oracle's restatement on (a) two further
independent restatements (grid_sample based; plain C) and (b) known-answer identities that fix tap
order, offset sign / (dy,dx) channel order, mask and deformable-group layout."""
import pytest
import torch
import torch.nn.functional as F

from oracle import c_ref
from oracle import eavsr_oracle as O
from tests import helpers as H
from tests.golden import cases


def _inputs(n=1, c=16, h=9, w=11, co=8, dg=2, sigma=2.0, seed=0):
    x = cases.randn(seed + 1, n, c, h, w)
    off = cases.randn(seed + 2, n, dg * 18, h, w, scale=sigma)
    mask = cases.rand(seed + 3, n, dg * 9, h, w)
    wt = cases.randn(seed + 4, co, c, 3, 3, scale=0.1)
    b = cases.randn(seed + 5, co, scale=0.1)
    return x, off, mask, wt, b


@pytest.mark.parametrize("sigma", [0.3, 3.0, 9.0])
def test_three_restatements_agree(sigma):
    x, off, mask, wt, b = _inputs(sigma=sigma)
    a = O.dcnv2(x, off, mask, wt, b, 1, 1, 1, 1, 2)
    g = O.dcnv2_via_grid_sample(x, off, mask, wt, b, 1, 1, 1, 1, 2)
    c = c_ref.dcnv2(x, off, mask, wt, b, 1, 1, 1, 2)
    assert H.maxabs(a, g) <= 2e-5
    assert H.maxabs(a, c) <= 2e-5


def test_zero_offset_unit_mask_is_conv2d():
    x, off, mask, wt, b = _inputs()
    out = O.dcnv2(x, off * 0, torch.ones_like(mask), wt, b, 1, 1, 1, 1, 2)
    assert H.maxabs(out, F.conv2d(x, wt, b, 1, 1)) <= 1e-5


def _shifted_conv(x, wt, b, dy, dx, m=4):
    """conv2d sampling the zero-extended image at (y - 1 + i + dy, x - 1 + j + dx), integer dy, dx"""
    n, c, h, w = x.shape
    xs = torch.zeros(n, c, h + 2 * m, w + 2 * m)
    xs[:, :, m:-m, m:-m] = x
    full = F.conv2d(xs, wt, b, 1, 1)
    return full[:, :, m + dy:m + dy + h, m + dx:m + dx + w]


def test_offset_channel_order_is_dy_then_dx():
    """channel 2k carries dy and 2k+1 carries dx of tap k (mmcv order, as AdaptBlockOffset emits it)"""
    x, off, mask, wt, b = _inputs(dg=1)
    for dy, dx in ((1, 0), (0, -1), (2, -3)):
        off = torch.zeros_like(off)
        off[:, 0::2] = float(dy)
        off[:, 1::2] = float(dx)
        out = O.dcnv2(x, off, torch.ones_like(mask), wt, b, 1, 1, 1, 1, 1)
        assert H.maxabs(out, _shifted_conv(x, wt, b, dy, dx)) <= 1e-5, (dy, dx)


def test_mask_and_group_layout():
    """mask channel g*9+k scales tap k of the channels of group g only"""
    x, off, mask, wt, b = _inputs(dg=2)
    m = torch.ones_like(mask)
    m[:, 9 + 4] = 0.0  # centre tap of group 1
    out = O.dcnv2(x, off * 0, m, wt, None, 1, 1, 1, 1, 2)
    w2 = wt.clone()
    w2[:, 8:, 1, 1] = 0.0
    assert H.maxabs(out, F.conv2d(x, w2, None, 1, 1)) <= 1e-5


def test_half_pixel_offset_is_the_mean_of_neighbours_and_border_is_zero_padded():
    x = torch.arange(12.0).view(1, 1, 3, 4).repeat(1, 8, 1, 1)
    wt = torch.zeros(1, 8, 3, 3)
    wt[0, 0, 1, 1] = 1.0  # picks the centre tap of channel 0
    off = torch.zeros(1, 18, 3, 4)
    off[:, 9] = 0.5  # dx of tap 4
    out = O.dcnv2(x, off, torch.ones(1, 9, 3, 4), wt, None, 1, 1, 1, 1, 1)
    ref = x[:, :1].clone()
    ref[..., :-1] = 0.5 * (x[:, :1, :, :-1] + x[:, :1, :, 1:])
    ref[..., -1] = 0.5 * x[:, :1, :, -1]  # right neighbour is outside: contributes zero
    assert H.maxabs(out, ref) <= 1e-6


def test_c_flow_warp_agrees_with_oracle():
    for name, (x, flow, pad) in cases.g1_flow_warp_cases().items():
        assert H.maxabs(c_ref.flow_warp(x, flow, pad), O.flow_warp(x, flow, pad)) <= 5e-5 * max(1, x.abs().max().item()), name


# ---- committed known answers (tests/golden/gen_known_answers.py) ---------------------------------------
@pytest.mark.parametrize("name", list(cases.G4_CASES))
def test_every_restatement_reproduces_the_committed_known_answers(name):
    """G4 was computed by the plain-C restatement with double accumulation; the two PyTorch restatements and a fresh
    run of the C code must reproduce the stored outputs -- borders, |offset| > 1, dg = 8, exact boundary positions."""
    gold = H.golden("g4_dcnv2")[name]
    x, off, mask, wt, b, dg = cases.g4_inputs(name)
    tol = 2e-5 * max(1.0, gold.abs().max().item())
    assert H.maxabs(c_ref.dcnv2(x, off, mask, wt, b, 1, 1, 1, dg), gold) == 0.0
    assert H.maxabs(O.dcnv2(x, off, mask, wt, b, 1, 1, 1, 1, dg), gold) <= tol
    if name != "dg8_edge":   # grid_sample's normalise / un-normalise round trip moves exact-boundary positions by an ulp
        assert H.maxabs(O.dcnv2_via_grid_sample(x, off, mask, wt, b, 1, 1, 1, 1, dg), gold) <= tol


@pytest.mark.parametrize("name", ["dg8_s2", "dg2_s2_c16", "dg1_s1p5_ragged"])
def test_oracle_autograd_reproduces_the_committed_gradients(name):
    """G9: fp64 autograd gradients, stored as fp32; the fp32 oracle the GPU tests differentiate must agree with them."""
    gold = H.golden("g9_gradients")
    x, off, mask, wt, b, dg = cases.g4_inputs(name)
    leaves = [t.clone().requires_grad_(True) for t in (x, off, mask, wt, b)]
    out = O.dcnv2(*leaves, 1, 1, 1, 1, dg)
    grads = torch.autograd.grad((out * cases.g9_cotangent(name, out.shape)).sum(), leaves)
    for key, gr in zip(("dx", "doffset", "dmask", "dweight", "dbias"), grads):
        want = gold[f"dcn_{name}__{key}"]
        assert H.maxabs(gr, want) <= 1e-4 * max(1.0, want.abs().max().item()), key
    for wname, (xw, flow, pad) in cases.g1_flow_warp_cases().items():
        lv = [xw.clone().requires_grad_(True), flow.clone().requires_grad_(True)]
        o = O.flow_warp(lv[0], lv[1], pad)
        dx, dflow = torch.autograd.grad((o * cases.g9_cotangent(wname, o.shape)).sum(), lv)
        for key, gr in (("dx", dx), ("dflow", dflow)):
            if (wname, key) == ("c4_int_zeros", "dflow"):
                continue    # exact-integer positions: d/dflow is discontinuous there and the side taken depends on the
                            # last ulp of the normalise / un-normalise round trip (fp32 vs the fixture's fp64)
            want = gold[f"warp_{wname}__{key}"]
            bad = ((gr - want).abs() > 2e-4 * max(1.0, want.abs().max().item())).float().mean().item()
            assert bad <= 0.002, (wname, key, bad)     # fp32 vs fp64 floor() may differ at a handful of positions
