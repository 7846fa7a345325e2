"""GPU parity tests, op level: every HIP kernel behind the C ABI against the CPU oracle and the
golden vectors.  Run with `pytest -m gpu` on a MI355X."""
import pytest
import torch
import torch.nn.functional as F

from oracle import eavsr_oracle as O
from tests import helpers as H
from tests.golden import cases

pytestmark = pytest.mark.gpu

DEFAULT_CONV_MODE = "winograd4"   # eavsr_amd.ops.CONV_MODE's default; tests that switch modes restore it
DEFAULT_DCN_MODE = "il6"           # eavsr_amd.ops.DCN_MODE's default


@pytest.fixture(scope="module")
def ops(cuda):
    from eavsr_amd import ops as _ops
    _ops.lib()  # fails loudly if the HIP extension is missing
    return _ops


def g(t, dev):
    return t.to(dev).contiguous()


def rel_err(a, b):
    return H.maxabs(a, b) / max(1e-6, b.abs().max().item())


def test_mfma_layout_selftest(ops):
    assert ops.selftest_mfma("cuda:0") == 0


# ------------------------------------------------------------------------------------------ a1/a2
def test_flow_warp_golden(ops, cuda):
    gold = H.golden("g1_flow_warp")
    for name, (x, flow, pad) in cases.g1_flow_warp_cases().items():
        a = ops.flow_warp(g(x, cuda), g(flow, cuda), padding_mode=pad).cpu()
        assert H.maxabs(a, gold[name + "__nchw"]) <= 2e-5 * max(1.0, x.abs().max().item()), name
        b = ops.flow_warp(g(x, cuda), g(flow.permute(0, 2, 3, 1), cuda), padding_mode=pad, flow_layout="nhwc").cpu()
        assert H.maxabs(b, gold[name + "__nhwc"]) <= 2e-5 * max(1.0, x.abs().max().item()), name
        # a permuted NCHW view is accepted without a copy
        c = ops.flow_warp(g(x, cuda), g(flow, cuda).permute(0, 2, 3, 1), padding_mode=pad, flow_layout="nhwc").cpu()
        assert torch.equal(a, c), name


@pytest.mark.parametrize("shape", [(1, 64, 45, 80), (2, 64, 90, 160), (3, 7, 33, 70), (1, 1, 1, 1), (2, 3, 6, 10)])
@pytest.mark.parametrize("pad", ["zeros", "border"])
def test_flow_warp_vs_oracle(ops, cuda, shape, pad):
    n, c, h, w = shape
    x = cases.randn(1, n, c, h, w)
    f1 = cases.randn(2, n, 2, h, w, scale=2.5)
    f2 = cases.randn(3, n, 2, h, w, scale=1.0)
    ref = O.flow_warp(x, f1 + f2, pad)
    out = ops.flow_warp(g(x, cuda), g(f1, cuda), padding_mode=pad, flow2=g(f2, cuda)).cpu()
    assert H.maxabs(out, ref) <= 5e-5


def test_flow_warp_identity_and_integer_shift(ops, cuda):
    x = cases.randn(5, 2, 16, 20, 24)
    z = torch.zeros(2, 2, 20, 24)
    # not bit-exact by design: the kernel repeats the reference's normalise / un-normalise round trip
    assert H.maxabs(ops.flow_warp(g(x, cuda), g(z, cuda)).cpu(), x) <= 1e-5
    f = torch.zeros(2, 2, 20, 24)
    f[:, 0] = 3.0
    f[:, 1] = -2.0
    out = ops.flow_warp(g(x, cuda), g(f, cuda)).cpu()
    ref = torch.zeros_like(x)
    ref[:, :, 2:, :-3] = x[:, :, :-2, 3:]
    assert H.maxabs(out, ref) <= 1e-5


def test_flow_warp_rejects_bad_inputs(ops, cuda):
    x = torch.zeros(1, 2, 8, 8, device=cuda)
    with pytest.raises(ValueError):
        ops.flow_warp(x, torch.zeros(1, 2, 8, 9, device=cuda))
    with pytest.raises(ValueError):
        ops.flow_warp(x, torch.zeros(1, 2, 8, 8, device=cuda), padding_mode="mirror")
    with pytest.raises(ValueError):
        ops.flow_warp(x, torch.zeros(1, 2, 8, 8, device=cuda), interpolation="bicubic")
    with pytest.raises(RuntimeError):
        ops.flow_warp(torch.zeros(1, 2, 8, 8), torch.zeros(1, 2, 8, 8))  # CPU tensor: no CPU path


# ------------------------------------------------------------------------------------------ conv
CONV_CASES = [
    # k, srcs channels, cout, n, h, w, act, residual, partial
    (3, [64], 64, 2, 37, 45, "relu", False, False),
    (3, [64], 64, 1, 16, 32, None, True, True),
    (3, [64, 64, 64], 64, 1, 20, 40, "lrelu", False, False),
    (3, [64, 64, 64, 64, 64], 64, 1, 18, 33, "lrelu", False, False),
    (3, [18], 2, 2, 23, 31, None, False, False),
    (3, [3], 64, 1, 19, 21, "relu", False, False),
    (3, [64], 6, 1, 12, 16, None, False, False),
    (3, [64], 256, 1, 17, 35, "lrelu", False, False),
    (3, [256], 64, 1, 9, 40, None, False, False),
    (3, [64], 3, 1, 33, 65, None, True, False),
    (1, [64, 64, 64], 64, 2, 21, 37, None, False, False),
    (5, [64], 120, 1, 22, 36, None, False, False),
    (7, [8], 32, 2, 12, 20, "relu", False, False),
    (7, [32], 64, 1, 24, 40, "relu", False, False),
    (7, [64], 32, 1, 6, 10, "relu", False, False),
    (7, [16], 2, 1, 48, 80, None, False, False),
    # the host picks the tile height from the tile count (eavsr_conv2d_tile_rows): everything above runs 8-row
    # tiles; these run 16-row (4 x 9 x 5 tiles) and 32-row (10 x 5 x 5 tiles) tiles
    (3, [64], 64, 4, 133, 156, "relu", True, True),
    (3, [64, 64], 64, 10, 133, 156, "lrelu", True, True),
    (5, [64], 120, 4, 133, 156, None, False, False),
    (1, [64, 64, 64], 64, 10, 133, 156, None, False, False),
    (3, [18], 2, 10, 130, 155, None, False, False),
]


@pytest.fixture()
def direct_conv(ops):
    """The direct fp32-MFMA convolution kernel (the default mode sends large 3x3 problems to the Winograd kernel)."""
    ops.set_conv_mode("direct")
    yield ops
    ops.set_conv_mode(DEFAULT_CONV_MODE)


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: f"k{c[0]}_c{'+'.join(map(str, c[1]))}_o{c[2]}_{c[4]}x{c[5]}")
def test_conv2d_vs_torch_cpu(ops, cuda, case, direct_conv):
    k, chans, cout, n, h, w, act, use_res, use_part = case
    if h > 100:   # the large cases exist to cover the taller tiles
        assert ops.lib().eavsr_conv2d_tile_rows(n, h, w, k) == (16 if n == 4 else 32)
    cin = sum(chans)
    srcs = [cases.randn(10 + i, n, c, h, w) for i, c in enumerate(chans)]
    wt = cases.randn(20, cout, cin, k, k, scale=1.0 / (cin * k * k) ** 0.5)
    b = cases.randn(21, cout, scale=0.1)
    res = cases.randn(22, n, cout, h, w) if use_res else None
    ref = F.conv2d(torch.cat(srcs, 1), wt, b, 1, k // 2)
    if act == "relu":
        ref = F.relu(ref)
    elif act == "lrelu":
        ref = F.leaky_relu(ref, 0.1)
    pre = ref
    if use_res:
        ref = ref + res
    out = ops.conv2d([g(s, cuda) for s in srcs], g(wt, cuda), g(b, cuda), act=act, slope=0.1,
                     residual=None if res is None else g(res, cuda), chan_partial=use_part)
    if use_part:
        out, part = out
        sums = part.sum(dim=1).cpu()
        assert H.maxabs(sums, pre.sum(dim=(2, 3))) <= 2e-3
    assert H.maxabs(out.cpu(), ref) <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("case", [([64], 6, 2, 45, 80, "lrelu", False), ([18], 2, 2, 24, 32, None, False), ([64], 3, 1, 33, 64, None, True),
                                  ([64], 4, 1, 10, 72, "relu", False), ([18], 2, 3, 180, 320, None, False),
                                  ([20], 3, 1, 9, 136, None, True)])
def test_conv3x3_small_cout_co_resident_variant(ops, cuda, case):
    """round 3: conv3x3_smallco_lite (256 threads, <= 40 registers, scalar weights from the packed form; the variant that fits beside
    a resident Winograd workgroup) against torch CPU and against the classic kernel, incl. a ragged last stage (18 and 20 input
    channels over 4-channel stages), tiles cut by the image, several sources, residual."""
    chans, cout, n, h, w, act, use_res = case
    cin = sum(chans)
    srcs = [cases.randn(30 + i, n, c, h, w) for i, c in enumerate(chans)]
    wt = cases.randn(40, cout, cin, 3, 3, scale=1.0 / (cin * 9) ** 0.5)
    b = cases.randn(41, cout, scale=0.1)
    res = cases.randn(42, n, cout, h, w) if use_res else None
    ref = F.conv2d(torch.cat(srcs, 1), wt, b, 1, 1)
    ref = F.relu(ref) if act == "relu" else F.leaky_relu(ref, 0.1) if act == "lrelu" else ref
    if use_res:
        ref = ref + res
    outs = {}
    was = ops.SMALLCO_LITE, ops.SMALLCO_LITE_MIN_TILES
    try:
        ops.SMALLCO_LITE_MIN_TILES = 0          # (the default sends only launches of >= 1024 workgroups to the variant)
        for lite in (True, False):
            ops.SMALLCO_LITE = lite
            with ops.profile() as prof:
                outs[lite] = ops.conv2d([g(s_, cuda) for s_ in srcs], g(wt, cuda), g(b, cuda), act=act, slope=0.1,
                                        residual=None if res is None else g(res, cuda)).cpu()
            assert list(prof.summary()) == [f"conv3x3_{cin}to{cout}"]
    finally:
        ops.SMALLCO_LITE, ops.SMALLCO_LITE_MIN_TILES = was
    tol = 2e-5 * max(1.0, ref.abs().max().item())
    assert H.maxabs(outs[True], ref) <= tol and H.maxabs(outs[False], ref) <= tol
    assert H.maxabs(outs[True], outs[False]) <= tol


@pytest.mark.parametrize("case", [(8, 32, 2, 24, 40, "relu"), (32, 64, 1, 48, 80, "relu"), (64, 32, 2, 16, 32, "relu"), (32, 16, 3, 6, 10, "relu"),
                                  (16, 2, 2, 12, 20, None), (64, 64, 1, 37, 70, "lrelu"), (8, 70, 1, 19, 33, None), (24, 32, 1, 96, 160, "relu"),
                                  (8, 64, 24, 48, 80, "relu"), (8, 32, 24, 45, 80, "relu"), (8, 64, 24, 24, 40, "relu"), (16, 48, 13, 40, 70, None),
                                  (8, 96, 4, 70, 100, "relu")])
def test_conv7x7_bf16x6_matches_fp64_and_the_fp32_kernel(ops, cuda, case):
    """round 3: eavsr_conv7x7_f32x6 (SPyNet's basic module, eavsrp_model.py:398-431: the contraction on the bf16 matrix pipe, both
    operands split exactly into three bf16 terms) at every layer shape of the module, tiles cut by the image in both directions, a
    cout that is not a multiple of 32, three input chunks, launches of every workgroup shape the entry point chooses between (16- or
    8-row tiles, one or two 32-channel tiles per workgroup, two workgroups per packed 64-channel block), against an fp64 evaluation (as accurate as the fp32-MFMA kernel, whose
    error is measured beside it) and against torch's fp32 CPU convolution."""
    cin, cout, n, h, w, act = case
    x = cases.randn(50, n, cin, h, w)
    wt = cases.randn(51, cout, cin, 7, 7, scale=1.0 / (cin * 49) ** 0.5)
    b = cases.randn(52, cout, scale=0.1)
    ref64 = F.conv2d(x.double(), wt.double(), b.double(), 1, 3)
    ref64 = F.relu(ref64) if act == "relu" else F.leaky_relu(ref64, 0.1) if act == "lrelu" else ref64
    outs = {}
    was = ops.CONV7_MODE
    try:
        for mode in ("bf16x6", "fp32"):
            ops.CONV7_MODE = mode
            with ops.profile() as prof:
                outs[mode] = ops.conv2d(g(x, cuda), g(wt, cuda), g(b, cuda), act=act, slope=0.1).cpu()
            assert list(prof.summary()) == [f"conv7x7_{cin}to{cout}_x6" if mode == "bf16x6" else f"conv7x7_{cin}to{cout}"]
    finally:
        ops.CONV7_MODE = was
    scale = max(1.0, ref64.abs().max().item())
    e6 = (outs["bf16x6"].double() - ref64).abs().max().item() / scale
    e32 = (outs["fp32"].double() - ref64).abs().max().item() / scale
    assert e6 <= 3e-6 and e6 <= 2.0 * e32 + 2e-7, (e6, e32)
    ref = ref64.float()
    assert H.maxabs(outs["bf16x6"], ref) <= 2e-5 * scale


@pytest.mark.parametrize("case", [(64, (32, 16, 72), 2, 45, 80), (64, (32, 16, 72), 1, 180, 320), (64, (8, 4, 18), 3, 23, 40), (8, (40,), 24, 24, 40),
                                  (16, (70,), 30, 30, 50)])
def test_conv5x5_bf16x6_matches_fp64_and_winograd(ops, cuda, case):
    """round 3: the predictor's 5x5 heads (networks.py:289-315, three weights in one launch) by eavsr_conv_f32x6 against an fp64
    evaluation and the F(2x2,5x5) fp32 kernel it replaces; the last slab of a chunk has 6 k-steps, the zero tap is tap 25."""
    cin, couts, n, h, w = case
    x = cases.randn(60, n, cin, h, w)
    ws = [cases.randn(61 + i, co, cin, 5, 5, scale=1.0 / (cin * 25) ** 0.5) for i, co in enumerate(couts)]
    bs = [cases.randn(65 + i, co, scale=0.1) for i, co in enumerate(couts)]
    ref64 = F.conv2d(x.double(), torch.cat(ws).double(), torch.cat(bs).double(), 1, 2)
    outs = {}
    was = ops.CONV5_MODE
    try:
        for mode in ("bf16x6", "wino"):
            ops.CONV5_MODE = mode
            with ops.profile() as prof:
                outs[mode] = ops.conv2d(g(x, cuda), [g(w_, cuda) for w_ in ws], [g(b_, cuda) for b_ in bs]).cpu()
            assert (list(prof.summary()) == [f"conv5x5_{cin}to{sum(couts)}_x6"]) == (mode == "bf16x6")
    finally:
        ops.CONV5_MODE = was
    scale = max(1.0, ref64.abs().max().item())
    e6 = (outs["bf16x6"].double() - ref64).abs().max().item() / scale
    ew = (outs["wino"].double() - ref64).abs().max().item() / scale
    assert e6 <= 3e-6 and e6 <= 2.0 * ew + 2e-7, (e6, ew)


@pytest.mark.parametrize("case", [(64, 2, 96, 96, "relu", None), (64, 2, 96, 96, None, "sums"), (64, 2, 96, 96, None, "residual"),
                                  (64, 2, 96, 96, "relu_mask", None), (64, 1, 45, 80, "lrelu", "residual"), (40, 3, 19, 37, "relu", "sums"),
                                  (32, 1, 8, 32, None, None), (70, 1, 1, 1, None, "residual"), (64, 2, 3, 2, "relu", "sums"),
                                  (64, 1, 23, 200, "relu_mask", None), (6, 1, 30, 50, None, None), (64, 0, 9, 9, None, None)])
def test_conv3x3_small_launches_bf16x6_matches_fp64_and_the_fp32_kernel(ops, cuda, case):
    """round 5: eavsr_conv3x3_f32x6s -- RCABlock's convolutions and their input-gradient convolutions (networks.py:456-464) at a
    training crop and below: every epilogue of the descriptor (bias, ReLU / LReLU, residual, EAVSR_ACT_RELU_MASK, per-tile channel
    sums), tiles cut by the image, couts that are not multiples of 32, a single pixel, an empty batch -- against an fp64 evaluation
    (as accurate as the fp32-MFMA kernel, whose error is measured beside it)."""
    cout, n, h, w, act, extra = case
    x = cases.randn(90, n, 64, h, w)
    wt = cases.randn(91, cout, 64, 3, 3, scale=1.0 / 24)
    b = cases.randn(92, cout, scale=0.1)
    r = torch.relu(cases.randn(93, n, cout, h, w)) if (extra == "residual" or act == "relu_mask") else None
    ref64 = F.conv2d(x.double(), wt.double(), b.double(), 1, 1)
    ref64 = F.relu(ref64) if act == "relu" else F.leaky_relu(ref64, 0.1) if act == "lrelu" else ref64
    sums64 = ref64.sum((2, 3))
    if act == "relu_mask":
        ref64 = torch.where(r > 0, ref64, torch.zeros_like(ref64))
    elif r is not None:
        ref64 = ref64 + r.double()
    outs, sums = {}, {}
    was = ops.CONV3_SMALL
    try:
        for mode in ("x6s", "direct") if n else ("x6s",):
            ops.CONV3_SMALL = mode
            with ops.profile() as prof:
                y = ops.conv2d(g(x, cuda), g(wt, cuda), g(b, cuda), act=act, slope=0.1, residual=None if r is None else g(r, cuda),
                               chan_partial=extra == "sums")
            if extra == "sums":
                y, part = y
                sums[mode] = part.sum(1).cpu()
            outs[mode] = y.cpu()
            if n and cout not in (6,):
                assert (list(prof.summary()) == [f"conv3x3_64to{cout}_x6s"]) == (mode == "x6s"), list(prof.summary())
    finally:
        ops.CONV3_SMALL = was
    assert outs["x6s"].shape == ref64.shape
    if n == 0:
        return
    scale = max(1.0, ref64.abs().max().item())
    e6 = (outs["x6s"].double() - ref64).abs().max().item() / scale
    e32 = (outs["direct"].double() - ref64).abs().max().item() / scale
    assert e6 <= 3e-6 and e6 <= 2.0 * e32 + 2e-7, (e6, e32)
    if extra == "sums":
        assert (sums["x6s"].double() - sums64).abs().max().item() <= 2e-5 * max(1.0, sums64.abs().max().item())
    if act == "relu_mask":
        assert (outs["x6s"][r == 0] == 0).all()


@pytest.mark.parametrize("shape", [(2, 180, 320), (3, 133, 156), (8, 64, 128), (2, 3, 512), (1, 45, 80), (2, 19, 37), (1, 1, 8), (3, 5, 1)])
def test_rcab_attention_before_the_second_convolution_fp32(ops, cuda, shape):
    """eavsr_ca_scale_pre_f32 + the scaled-residual epilogue of the F(4x4,3x3) kernel (desc.res_scale): the attention of an RCAB
    (CALayer, networks.py:444-447) from border-corrected channel sums of the second convolution's INPUT must equal the attention
    from that convolution's OUTPUT, and `x + scale * conv(t)` as the convolution's epilogue the separate scale_residual launch --
    on a launch the Winograd kernel takes, on small / ragged ones (other kernels: the tail as its own launch), single rows / columns."""
    n, h, w = shape
    x = cases.randn(101, n, 64, h, w)
    w1, w2 = cases.randn(102, 64, 64, 3, 3, scale=1.0 / 24), cases.randn(103, 64, 64, 3, 3, scale=1.0 / 24)
    b1, b2 = cases.randn(104, 64, scale=0.1), cases.randn(105, 64, scale=0.1)
    a_w, a_b, c_w, c_b = cases.randn(106, 4, 64, scale=0.5), cases.randn(107, 4, scale=0.1), cases.randn(108, 64, 4, scale=0.5), cases.randn(109, 64, scale=0.1)
    gx, gw1, gw2, gb1, gb2 = (g(v, cuda) for v in (x, w1, w2, b1, b2))
    ga_w, ga_b, gc_w, gc_b = (g(v, cuda) for v in (a_w, a_b, c_w, c_b))
    # reference (fp64): the block as the reference computes it
    t64 = F.relu(F.conv2d(x.double(), w1.double(), b1.double(), 1, 1))
    r64 = F.conv2d(t64, w2.double(), b2.double(), 1, 1)
    m64 = r64.mean((2, 3))
    s64 = torch.sigmoid(F.relu(m64 @ a_w.double().t() + a_b.double()) @ c_w.double().t() + c_b.double())
    y64 = x.double() + r64 * s64[:, :, None, None]
    t, tpart = ops.conv2d(gx, gw1, gb1, act="relu", chan_partial=True)
    scale = ops.ca_scale_pre(t, tpart, gw2, gb2, ga_w, ga_b, gc_w, gc_b)
    assert H.maxabs(scale.cpu().double(), s64) <= 2e-5
    # round 6: the border-line sums as a by-product of the first convolution's epilogue (desc.border_pieces) instead of a launch of
    # their own -- where the grouped F(4x4,3x3) kernel runs (None on every other route): same t, same plane sums, the same attention
    # to summation order, and the pieces themselves equal the border lines of t (ragged heights / widths, single tile rows)
    with ops.profile() as prof:
        t2, tpart2, pieces = ops.conv2d(gx, gw1, gb1, act="relu", chan_partial=True, border=True)
    assert torch.equal(t2, t) and torch.equal(tpart2, tpart)
    if "conv3x3_64to64_wino4" in set(prof.summary()):
        assert pieces is not None and (n, h, w) not in ((1, 45, 80), (2, 19, 37))
        pd = pieces.data.cpu()
        tc = t.cpu()
        lines = [tc[:, :, 0, :].sum(-1), tc[:, :, h - 1, :].sum(-1), tc[:, :, :, 0].sum(-1), tc[:, :, :, w - 1].sum(-1)]
        for bi, (want, cnt) in enumerate(zip(lines, (pieces.p_rows, pieces.p_rows, pieces.p_cols, pieces.p_cols))):
            got = pd[:, bi, :cnt].sum(1)
            assert H.maxabs(got, want) <= 1e-5 * max(1.0, want.abs().max().item()), (bi, H.maxabs(got, want))
        with ops.profile() as prof2:
            scale_p = ops.ca_scale_pre(t, tpart, gw2, gb2, ga_w, ga_b, gc_w, gc_b, border=pieces)
        assert prof2.summary()["ca_scale_pre"]["calls"] == 1
        assert H.maxabs(scale_p.cpu(), scale.cpu()) <= 2e-6
        assert H.maxabs(scale_p.cpu().double(), s64) <= 2e-5
    else:
        assert pieces is None
    with ops.profile() as prof:
        y = ops.conv2d(t, gw2, gb2, residual=gx, res_scale=scale)
    if (n, h, w) in ((2, 180, 320), (3, 133, 156), (8, 64, 128)):
        assert list(prof.summary()) == ["conv3x3_64to64_wino4"]      # the epilogue form: one launch
    assert H.maxabs(y.cpu().double(), y64) <= 1e-4 * max(1.0, y64.abs().max().item())
    r, rpart = ops.conv2d(t, gw2, gb2, chan_partial=True)
    y_old = ops.scale_residual(r, ops.ca_scale(rpart, h * w, ga_w, ga_b, gc_w, gc_b), gx)
    assert H.maxabs(y.cpu(), y_old.cpu()) <= 2e-5 * max(1.0, y64.abs().max().item())
    with pytest.raises(ValueError):
        ops.conv2d(t, gw2, gb2, res_scale=scale)      # no residual


@pytest.mark.parametrize("case", [(64, 2, 96, 96), (40, 1, 19, 37), (64, 1, 180, 320), (6, 1, 24, 40)])
def test_conv2d_dgrad_flag_reads_the_forward_weight_in_place(ops, cuda, case):
    """conv2d(dY, W, dgrad=True) = the input gradient of conv(x, W): the small-launch bf16x6 kernel packs the transposed, flipped
    weight straight from W (eavsr_pack_conv_weight_x6_dgrad) -- bit-identical to the same kernel on the materialised copy; large
    launches and small channel counts materialise it (the other kernels' routes), same values as torch's conv_transpose."""
    cin_w, n, h, w = case
    dy = cases.randn(95, n, 64, h, w)
    wt = cases.randn(96, 64, cin_w, 3, 3, scale=1.0 / 24)
    r = torch.relu(cases.randn(97, n, cin_w, h, w))
    ref = torch.nn.grad.conv2d_input((n, cin_w, h, w), wt.double(), dy.double(), padding=1)
    gd, gw, gr = g(dy, cuda), g(wt, cuda), g(r, cuda)
    for kw, want in ((dict(), ref), (dict(residual=gr), ref + r.double()),
                     (dict(residual=gr, act="relu_mask"), torch.where(r > 0, ref, torch.zeros_like(ref)))):
        got = ops.conv2d(gd, gw, None, dgrad=True, **kw)
        mat = ops.conv2d(gd, ops.dgrad_weight(gw), None, **kw)
        assert torch.equal(got, mat)
        assert H.maxabs(got.cpu().double(), want) <= 2e-5 * max(1.0, want.abs().max().item())
    with pytest.raises(ValueError):
        ops.conv2d(gd, gw, g(cases.randn(98, cin_w), cuda), dgrad=True)


@pytest.mark.parametrize("k", [5, 7])
@pytest.mark.parametrize("shape", [(8, 1, 1, 1, 1), (8, 33, 2, 3, 2), (16, 64, 1, 5, 200), (8, 5, 3, 2, 37), (40, 3, 0, 9, 9)])
def test_conv_bf16x6_degenerate_shapes(ops, cuda, k, shape):
    """images smaller than the kernel window (every tap but the centre ones in the zero padding), a single pixel, one output channel,
    an empty batch, a 200-pixel row of 5: eavsr_conv_f32x6 against torch's CPU convolution"""
    cin, cout, n, h, w = shape
    x = cases.randn(80, n, cin, h, w)
    wt = cases.randn(81, cout, cin, k, k, scale=1.0 / (cin * k * k) ** 0.5)
    b = cases.randn(82, cout, scale=0.1)
    out = ops.conv2d(g(x, cuda), g(wt, cuda), g(b, cuda)).cpu()
    ref = F.conv2d(x, wt, b, 1, k // 2)
    assert out.shape == ref.shape
    if n:
        assert H.maxabs(out, ref) <= 2e-5 * max(1.0, ref.abs().max().item())


def test_conv7x7_bf16x6_zero_tap_reads_no_neighbour(ops, cuda):
    """the 50th tap of a chunk has zero weights and reads the always-zero patch column: a NaN / Inf next to the receptive field of a
    pixel must not reach it (0 x NaN), and one inside must."""
    x = cases.randn(53, 1, 8, 20, 48)
    x[0, 3, 9, 30] = float("inf")
    wt = cases.randn(54, 32, 8, 7, 7, scale=0.05)
    out = ops.conv2d(g(x, cuda), g(wt, cuda), None).cpu()
    ref = F.conv2d(x, wt, None, 1, 3)
    fin = torch.isfinite(ref)
    assert torch.equal(torch.isfinite(out), fin)
    assert H.maxabs(out[fin], ref[fin]) <= 2e-5 * max(1.0, ref[fin].abs().max().item())


@pytest.mark.parametrize("mode", ["winograd4", "winograd", "direct", "bf16x9"])
def test_conv_epilogue_relu_of_minus_infinity_and_of_negatives_is_plus_zero(ops, cuda, mode):
    """the branch-free activation max(v, v * s) gave ReLU(-inf) = -inf (-inf * 0 = NaN, max returns the other operand)
    and ReLU(negative) = -0.0.  With the legacy product both are +0, as torch.relu gives; leaky / identity keep -inf."""
    x = cases.randn(71, 2, 64, 96, 128)
    wt = cases.randn(72, 64, 64, 3, 3, scale=0.05)
    b = cases.randn(73, 64, scale=0.1)
    b[5] = float("-inf")                                  # channel 5: -inf before the activation
    ops.set_conv_mode(mode)
    try:
        y = ops.conv2d(g(x, cuda), g(wt, cuda), g(b, cuda), act="relu").cpu()
        yl = ops.conv2d(g(x, cuda), g(wt, cuda), g(b, cuda), act="lrelu", slope=0.1).cpu()
    finally:
        ops.set_conv_mode(DEFAULT_CONV_MODE)
    ref = F.relu(F.conv2d(x, wt, b, 1, 1))
    assert torch.isfinite(y).all() and (y[:, 5] == 0).all()
    assert not torch.signbit(y[y == 0]).any()              # no -0.0
    assert H.maxabs(y, ref) <= 1e-4 * max(1.0, ref.abs().max().item())
    assert torch.isinf(yl[:, 5]).all() and (yl[:, 5] < 0).all()
    # the small-cout and the bf16x6 kernels share the epilogue form
    w3 = cases.randn(74, 3, 64, 3, 3, scale=0.05)
    b3 = torch.tensor([0.1, float("-inf"), -0.2])
    y3 = ops.conv2d(g(x, cuda), g(w3, cuda), g(b3, cuda), act="relu").cpu()
    assert (y3[:, 1] == 0).all() and not torch.signbit(y3[y3 == 0]).any()
    w5 = cases.randn(75, 16, 64, 5, 5, scale=0.03)
    b5 = cases.randn(76, 16, scale=0.1)
    b5[2] = float("-inf")
    y5 = ops.conv2d(g(x, cuda), g(w5, cuda), g(b5, cuda), act="relu").cpu()
    assert (y5[:, 2] == 0).all() and not torch.signbit(y5[y5 == 0]).any()


@pytest.mark.parametrize("mode", ["winograd4", "winograd", "direct", "bf16x9"])
def test_conv_epilogue_relu_propagates_nan_like_torch(ops, cuda, mode):
    """v_max_f32 returns its other operand for a NaN, so max(v, v (*) 0) turned ReLU(NaN) into +0 where
    torch.relu (networks.py:149-150) carries it to the loss.  eavsr_act() selects v when v is unordered: a NaN bias makes every
    output of that channel NaN through every activation form and every route; the other channels are untouched.  A NaN INPUT pixel
    makes at least the outputs torch makes NaN (the Winograd transforms spread it over the pixel's 6 x 6 tiles)."""
    nan = float("nan")
    for (n, h, w) in ((2, 96, 128), (2, 96, 96)):      # the Winograd launch | a crop-sized launch (bf16x6 small kernel in the default mode)
        x = cases.randn(81, n, 64, h, w)
        wt = cases.randn(82, 64, 64, 3, 3, scale=0.05)
        b = cases.randn(83, 64, scale=0.1)
        b[7] = nan
        ops.set_conv_mode(mode)
        try:
            ys = {a: ops.conv2d(g(x, cuda), g(wt, cuda), g(b, cuda), act=a, slope=0.1).cpu() for a in ("relu", "lrelu", None)}
            xn = x.clone()
            xn[1, 3, 40, 50] = nan
            yx = ops.conv2d(g(xn, cuda), g(wt, cuda), g(cases.randn(83, 64, scale=0.1), cuda), act="relu").cpu()
        finally:
            ops.set_conv_mode(DEFAULT_CONV_MODE)
        ref = F.relu(F.conv2d(x, wt, b, 1, 1))
        for a, y in ys.items():
            assert torch.isnan(y[:, 7]).all(), (mode, a)
            keep = [c for c in range(64) if c != 7]
            assert torch.isfinite(y[:, keep]).all(), (mode, a)
        assert H.maxabs(ys["relu"][:, keep], ref[:, keep]) <= 1e-4 * max(1.0, ref[:, keep].abs().max().item())
        refx = F.relu(F.conv2d(xn, wt, None, 1, 1))
        assert torch.isnan(yx[torch.isnan(refx)]).all(), mode
        assert torch.isfinite(yx[0]).all(), mode                   # the other sample is untouched
    # the small-cout, 5x5 bf16x6 and 1x1 kernels share the epilogue
    x = cases.randn(84, 2, 64, 24, 40)
    for k, cout in ((3, 3), (5, 16), (1, 64)):
        wk = cases.randn(85, cout, 64, k, k, scale=0.03)
        bk = cases.randn(86, cout, scale=0.1)
        bk[1] = nan
        yk = ops.conv2d(g(x, cuda), g(wk, cuda), g(bk, cuda), act="relu").cpu()
        assert torch.isnan(yk[:, 1]).all() and torch.isfinite(yk[:, [0, 2]]).all(), (k, cout)


def test_winograd4_hand_counted_reads_equal_the_compilers(ops, cuda):
    """Toolchain guard: conv_wino6.hip reads its MFMA operands (and the grouped transform its
    patch) by `asm volatile` LDS reads with hand-counted `s_waitcnt`; that is correct only while the compiler keeps a register
    between a read and its wait where THIS toolchain keeps it.  eavsr_amd/lib/guard/libwino4_creads.so is the same source
    built with -DEAVSR_W4_COMPILER_READS (every read and wait the compiler's; keyed on sources, flags and `hipcc --version`,
    prebuilt by __graft_entry__.build()): both libraries at the bench's launch, 2 x 64 x 180 x 320, must agree BIT FOR BIT -- the
    plain grouped kernel with channel sums, the scaled-residual instantiation (the RCAB tail as the epilogue), the duty-pair
    schedule (an odd number of 4-channel chunks) and the F(2x2,5x5) instantiation."""
    import ctypes as C
    from eavsr_amd import build, _native as N
    assert build.hipcc_version() in ("unknown hipcc",) or "clang" in build.hipcc_version()
    guard = C.CDLL(build.build_guard("wino4_creads"))
    for fn in ("eavsr_conv3x3_wino4_f32", "eavsr_conv5x5_wino_f32"):
        getattr(guard, fn).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        getattr(guard, fn).restype = C.c_int
    assert guard.eavsr_abi_version() == N.ABI_VERSION
    n, h, w = 2, 180, 320
    gen = torch.Generator().manual_seed(90)
    rn = lambda *sh: torch.randn(*sh, generator=gen).to(cuda)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def both(entry, k, cin, cout, act, residual=False, res_scale=False, sums=False):
        x, wt, b = rn(n, cin, h, w), rn(cout, cin, k, k) * 0.05, rn(cout) * 0.1
        res = rn(n, cout, h, w) if residual else None
        rs = torch.rand(n, cout, generator=gen).to(cuda) if res_scale else None
        wu = ops._packed_wino([wt], kind="f4" if k == 3 else "f5")
        tiles = (ops.lib().eavsr_conv3x3_wino4_tiles if k == 3 else ops.lib().eavsr_conv5x5_wino_tiles)(h, w)
        outs = []
        for lib in (ops.lib(), guard):
            out = torch.full((n, cout, h, w), float("nan"), device=cuda)
            part = torch.full((n, tiles, cout), float("nan"), device=cuda) if sums else None
            d = N.ConvDesc()
            d.src[0] = x.data_ptr(); d.src_c[0] = cin; d.n_src = 1; d.ksize = k
            d.bias = b.data_ptr(); d.out = out.data_ptr()
            d.residual = res.data_ptr() if res is not None else None
            d.res_scale = rs.data_ptr() if rs is not None else None
            d.chan_partial = part.data_ptr() if part is not None else None
            d.n, d.h, d.w, d.cin, d.cout = n, h, w, cin, cout
            d.act = {"none": 0, "relu": 1, "lrelu": 2}[act]
            d.slope = 0.1
            rc = getattr(lib, entry)(C.byref(d), C.c_void_p(wu.data_ptr()), st)
            assert rc == 0, (entry, rc)
            torch.cuda.synchronize()
            outs.append((out, part))
        (o1, p1), (o2, p2) = outs
        assert torch.isfinite(o1).all()
        assert torch.equal(o1, o2), (entry, k, cin, cout, (o1 - o2).abs().max().item())
        assert p1 is None or torch.equal(p1, p2)
        return o1, x, wt, b

    o, x, wt, b = both("eavsr_conv3x3_wino4_f32", 3, 64, 64, "relu", sums=True)               # conv-1 of an RCAB
    ref = F.relu(F.conv2d(x[:1].cpu(), wt.cpu(), b.cpu(), 1, 1))
    assert H.maxabs(o[:1].cpu(), ref) <= 1e-4 * max(1.0, ref.abs().max().item())             # (and both are right)
    both("eavsr_conv3x3_wino4_f32", 3, 64, 64, "none", residual=True, res_scale=True)          # conv-2 + the tail as its epilogue
    both("eavsr_conv3x3_wino4_f32", 3, 12, 40, "lrelu", residual=True)                         # 3 chunks: the duty-pair schedule
    both("eavsr_conv5x5_wino_f32", 5, 64, 120, "none")                                         # F(2x2,5x5)


def test_conv2d_ca_out_without_ca_raises_on_every_route(ops, cuda):
    """the bf16x6 route returned before the argument check"""
    x = torch.zeros(1, 64, 16, 16, device=cuda)
    for k in (3, 5, 7):
        with pytest.raises(ValueError):
            ops.conv2d(x, torch.zeros(64, 64, k, k, device=cuda), None, ca_out=True)


def test_pack_smallco_weight_rejects_output_counts_without_a_kernel(ops, cuda):
    """cout 1 / 5 were packed as 2 / 6 and the packing kernel read past the weight"""
    import ctypes as C
    lib = ops.lib()
    for cout in (1, 5, 7):
        assert lib.eavsr_smallco_packed_elems(cout, 64) == 0
        wt = torch.zeros(cout, 64, 3, 3, device=cuda)
        buf = torch.zeros(4096, device=cuda)
        assert lib.eavsr_pack_smallco_weight(C.c_void_p(wt.data_ptr()), C.c_void_p(buf.data_ptr()), cout, 64, None) == -2
    assert lib.eavsr_smallco_packed_elems(3, 64) > 0


def test_conv2d_multi_head_weights(ops, cuda):
    x = cases.randn(1, 1, 64, 14, 18)
    ws = [cases.randn(2 + i, co, 64, 3, 3, scale=0.05) for i, co in enumerate((4, 2))]
    bs = [cases.randn(5 + i, co, scale=0.1) for i, co in enumerate((4, 2))]
    out = ops.conv2d(g(x, cuda), [g(w_, cuda) for w_ in ws], [g(b_, cuda) for b_ in bs]).cpu()
    ref = F.conv2d(x, torch.cat(ws), torch.cat(bs), 1, 1)
    assert H.maxabs(out, ref) <= 2e-5


def test_conv2d_weight_cache_tracks_inplace_updates(ops, cuda):
    x = g(cases.randn(1, 1, 64, 16, 16), cuda)
    w_ = torch.nn.Parameter(g(cases.randn(2, 64, 64, 3, 3, scale=0.05), cuda))
    a = ops.conv2d(x, w_, None).cpu()
    with torch.no_grad():
        w_.mul_(2.0)
    b = ops.conv2d(x, w_, None).cpu()
    assert H.maxabs(b, 2 * a) <= 1e-5


@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("interp", ["bilinear", "nearest"])
@pytest.mark.parametrize("pad", ["zeros", "border", "reflection"])
def test_flow_warp_all_grid_sample_modes(ops, cuda, pad, interp, align):
    """networks.flow_warp forwards interpolation / padding_mode / align_corners to grid_sample (networks.py:733-738);
    the path uses (bilinear, zeros|border, True), the drop-in accepts the rest of the signature too."""
    from eavsr_amd import networks, eavsrp_model
    x = cases.randn(1, 2, 5, 19, 27)
    flow = cases.randn(2, 2, 2, 19, 27, scale=4.0)
    flow[:, :, :3] += 30.0                      # far outside: padding really matters
    # positions within 1e-3 of a rounding / cell boundary may legitimately flip under fp32 coordinate rounding
    ref = O.flow_warp(x, flow, pad, interp, align)
    got = networks.flow_warp(g(x, cuda), g(flow, cuda), interp, pad, align).cpu()
    got2 = eavsrp_model.flow_warp(g(x, cuda), g(flow.permute(0, 2, 3, 1).contiguous(), cuda), interp, pad, align).cpu()
    assert torch.equal(got, got2)
    bad = ((got - ref).abs() > 1e-4).float().mean().item()
    assert bad <= (0.002 if interp == "nearest" else 0.0), bad


@pytest.mark.parametrize("shape", [(2, 64, 24, 40), (1, 64, 13, 37), (1, 8, 9, 70), (3, 24, 5, 3)])
def test_flow_warp_pair_equals_two_single_warps(ops, cuda, shape):
    """networks.py:621 + :623 as one launch: the NCHW outputs bit-identical to the single warps, second output optionally IL8"""
    n, c, h, w = shape
    xa, xb = cases.randn(1, n, c, h, w), cases.randn(2, n, c, h, w)
    f1, f2 = cases.randn(3, n, 2, h, w, scale=3.0), cases.randn(4, n, 2, h, w)
    for flow2 in (None, f2):
        a1 = ops.flow_warp(g(xa, cuda), g(f1, cuda), flow2=None if flow2 is None else g(flow2, cuda))
        b1 = ops.flow_warp(g(xb, cuda), g(f1, cuda), flow2=None if flow2 is None else g(flow2, cuda))
        a2, b2 = ops.flow_warp_pair(g(xa, cuda), g(xb, cuda), g(f1, cuda), None if flow2 is None else g(flow2, cuda))
        assert torch.equal(a1, a2) and torch.equal(b1, b2)
        a3, b3 = ops.flow_warp_pair(g(xa, cuda), g(xb, cuda), g(f1, cuda), None if flow2 is None else g(flow2, cuda), b_il8=True)
        assert torch.equal(a1, a3)
        # same samples; the octet loop may contract multiply-adds differently from the plane loop (one rounding)
        assert H.maxabs(ops.to_il8(b1).cpu(), b3.cpu()) <= 1e-6 * max(1.0, xb.abs().max().item())
    assert H.maxabs(a1.cpu(), O.flow_warp(xa, f1 + f2)) <= 2e-5 * max(1.0, xa.abs().max().item())


# ------------------------------------------------------------------------------------------ a7
def _dcn_inputs(n, c, h, w, cout, dg, sigma, seed=0):
    x = cases.randn(seed + 1, n, c, h, w)
    off = cases.randn(seed + 2, n, dg * 18, h, w, scale=sigma)
    mask = cases.rand(seed + 3, n, dg * 9, h, w)
    wt = cases.randn(seed + 4, cout, c, 3, 3, scale=1.0 / (c * 9) ** 0.5)
    b = cases.randn(seed + 5, cout, scale=0.1)
    return x, off, mask, wt, b


@pytest.mark.parametrize("sigma", [0.0, 0.5, 2.0, 8.0])
@pytest.mark.parametrize("shape", [(1, 64, 24, 40, 64, 8), (2, 64, 13, 37, 64, 8), (1, 64, 10, 12, 64, 1),
                                   (1, 16, 9, 33, 32, 2), (1, 64, 7, 5, 40, 8),
                                   (1, 48, 12, 20, 32, 2), (1, 96, 8, 16, 16, 4)])     # 24 channels per group: 3 octets, not a
def test_dcnv2_vs_oracle(ops, cuda, shape, sigma):                                     # power of two -> the NCHW kernel
    n, c, h, w, cout, dg = shape
    x, off, mask, wt, b = _dcn_inputs(n, c, h, w, cout, dg, sigma)
    ref = O.dcnv2(x, off, mask, wt, b, 1, 1, 1, 1, dg)
    out = ops.modulated_deform_conv2d(g(x, cuda), g(off, cuda), g(mask, cuda), g(wt, cuda), g(b, cuda), 1, 1, 1, 1, dg)
    assert H.maxabs(out.cpu(), ref) <= 3e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("mode", ["native", "bf16x9", "il6", "il9"])
@pytest.mark.parametrize("name", list(cases.G4_CASES))
def test_dcnv2_committed_known_answers(ops, cuda, name, mode):
    """the stored outputs of the plain-C restatement (tests/golden/g4_dcnv2.npz) -- borders, |offset| > 1,
    dg = 8, sigma in {0.5, 2, 8}, exact validity-boundary positions -- through every DCNv2 kernel variant."""
    gold = H.golden("g4_dcnv2")[name]
    x, off, mask, wt, b, dg = cases.g4_inputs(name)
    prev = ops.DCN_MODE
    ops.set_dcn_mode(mode)
    try:
        out = ops.modulated_deform_conv2d(g(x, cuda), g(off, cuda), g(mask, cuda), g(wt, cuda), g(b, cuda), 1, 1, 1, 1, dg)
    finally:
        ops.set_dcn_mode(prev)
    assert H.maxabs(out.cpu(), gold) <= 3e-5 * max(1.0, gold.abs().max().item())


def test_dcnv2_zero_offset_unit_mask_is_conv2d(ops, cuda):
    x, off, mask, wt, b = _dcn_inputs(1, 64, 20, 30, 64, 8, 0.0)
    out = ops.modulated_deform_conv2d(g(x, cuda), g(off * 0, cuda), g(torch.ones_like(mask), cuda), g(wt, cuda),
                                      g(b, cuda), 1, 1, 1, 1, 8).cpu()
    assert H.maxabs(out, F.conv2d(x, wt, b, 1, 1)) <= 2e-5


def test_dcnv2_integer_offsets_shift_the_taps(ops, cuda):
    """offset (dy, dx) = (+1, -2) for every tap == conv2d of the image shifted by (-1, +2) with zero fill."""
    x, off, mask, wt, b = _dcn_inputs(1, 64, 12, 16, 64, 8, 0.0)
    off = torch.zeros_like(off)
    off[:, 0::2] = 1.0
    off[:, 1::2] = -2.0
    out = ops.modulated_deform_conv2d(g(x, cuda), g(off, cuda), g(torch.ones_like(mask), cuda), g(wt, cuda),
                                      g(b, cuda), 1, 1, 1, 1, 8).cpu()
    xs = torch.zeros(1, 64, 12 + 8, 16 + 8)
    xs[:, :, 4:-4, 4:-4] = x
    # sample at (y-1+i+1, x-1+j-2): correlate the zero-extended image, then crop with the shift
    full = F.conv2d(xs, wt, b, 1, 1)
    ref = full[:, :, 4 + 1:4 + 1 + 12, 4 - 2:4 - 2 + 16]
    assert H.maxabs(out, ref) <= 2e-5


@pytest.mark.parametrize("cfg", [
    # cin, cout, kh, kw, stride, pad, dil, groups, dg, h, w
    (12, 8, 3, 3, 1, 1, 1, 1, 3, 9, 11),        # 4 channels per deformable group (MultiAdSTN's default dg = 64 has 1)
    (64, 64, 3, 3, 1, 1, 1, 1, 64, 8, 10),      # deformable_groups = 64 (networks.py:576 default)
    (16, 24, 3, 3, 2, 1, 1, 1, 2, 13, 17),      # stride 2
    (8, 16, 5, 3, 1, 2, 1, 1, 1, 10, 12),       # 5 x 3 kernel, asymmetric output size
    (8, 8, 3, 3, 1, 2, 2, 1, 2, 11, 9),         # dilation 2
    (16, 16, 3, 3, 1, 1, 1, 2, 4, 7, 9),        # conv groups 2
], ids=lambda c: "c%d_o%d_k%dx%d_s%d_p%d_d%d_g%d_dg%d" % c[:9])
def test_dcnv2_rest_of_the_mmcv_signature(ops, cuda, cfg):
    """Configurations the reference never runs go through the generic kernel (forward only) and match the oracle."""
    cin, cout, kh, kw, st, pd, dl, groups, dg, h, w = cfg
    K = kh * kw
    ho = (h + 2 * pd - (dl * (kh - 1) + 1)) // st + 1
    wo = (w + 2 * pd - (dl * (kw - 1) + 1)) // st + 1
    x = cases.randn(1, 2, cin, h, w)
    off = cases.randn(2, 2, dg * 2 * K, ho, wo, scale=1.5)
    mask = cases.rand(3, 2, dg * K, ho, wo)
    wt = cases.randn(4, cout, cin // groups, kh, kw, scale=0.2)
    b = cases.randn(5, cout, scale=0.1)
    if groups == 1:
        ref = O.dcnv2(x, off, mask, wt, b, st, pd, dl, 1, dg)
    else:   # conv groups: each group of output channels sees its slice of the (already deformably sampled) input
        cig, cog, dgg = cin // groups, cout // groups, dg // groups
        parts = []
        for gi in range(groups):
            parts.append(O.dcnv2(x[:, gi * cig:(gi + 1) * cig], off[:, gi * dgg * 2 * K:(gi + 1) * dgg * 2 * K],
                                 mask[:, gi * dgg * K:(gi + 1) * dgg * K], wt[gi * cog:(gi + 1) * cog],
                                 b[gi * cog:(gi + 1) * cog], st, pd, dl, 1, dgg))
        ref = torch.cat(parts, 1)
    with ops.profile() as prof:
        out = ops.modulated_deform_conv2d(g(x, cuda), g(off, cuda), g(mask, cuda), g(wt, cuda), g(b, cuda), st, pd, dl,
                                          groups, dg)
    assert list(prof.summary()) == ["dcnv2_generic"]
    assert H.maxabs(out.cpu(), ref) <= 3e-5 * max(1.0, ref.abs().max().item())


def test_dcnv2_bad_arguments_raise(ops, cuda):
    z = lambda *s_: torch.zeros(*s_, device=cuda)
    with pytest.raises(ValueError):      # offset for the wrong number of deformable groups
        ops.modulated_deform_conv2d(z(1, 64, 8, 8), z(1, 18, 8, 8), z(1, 72, 8, 8), z(64, 64, 3, 3), None, 1, 1, 1, 1, 8)
    with pytest.raises(ValueError):      # weight / input channel mismatch
        ops.modulated_deform_conv2d(z(1, 64, 8, 8), z(1, 144, 8, 8), z(1, 72, 8, 8), z(64, 32, 3, 3), None, 1, 1, 1, 1, 8)
    with pytest.raises(RuntimeError):    # no CPU path
        ops.modulated_deform_conv2d(torch.zeros(1, 64, 8, 8), torch.zeros(1, 144, 8, 8), torch.zeros(1, 72, 8, 8),
                                    torch.zeros(64, 64, 3, 3), None, 1, 1, 1, 1, 8)


# ---- 3x3 conv by Winograd F(2x2, 3x3) on the fp32 MFMA: same descriptor / tensors / epilogue as the direct kernel ---
@pytest.fixture()
def conv_wino(ops):
    ops.set_conv_mode("winograd")
    yield ops
    ops.set_conv_mode(DEFAULT_CONV_MODE)


@pytest.mark.parametrize("case", [([64], 64, "relu", True, True, 10, 133, 156), ([64, 64], 64, "lrelu", True, False, 10, 133, 156),
                                  ([64], 256, None, False, False, 4, 100, 128), ([64, 64, 64, 64, 64], 64, "lrelu", False, False, 3, 180, 320),
                                  ([8], 40, None, False, True, 13, 65, 68), ([128], 64, "relu", False, True, 1, 400, 320)],
                         ids=lambda c: f"c{'+'.join(map(str, c[0]))}_o{c[1]}_{c[5]}x{c[6]}x{c[7]}")
def test_conv3x3_winograd_vs_torch_cpu(conv_wino, cuda, case):
    chans, cout, act, use_res, use_part, n, h, w = case
    cin = sum(chans)
    srcs = [cases.randn(10 + i, n, c, h, w) for i, c in enumerate(chans)]
    wt = cases.randn(20, cout, cin, 3, 3, scale=1.0 / (cin * 9) ** 0.5)
    b = cases.randn(21, cout, scale=0.1)
    res = cases.randn(22, n, cout, h, w) if use_res else None
    ref = F.conv2d(torch.cat(srcs, 1), wt, b, 1, 1)
    ref = F.relu(ref) if act == "relu" else (F.leaky_relu(ref, 0.1) if act == "lrelu" else ref)
    pre = ref
    if use_res:
        ref = ref + res
    with conv_wino.profile() as prof:
        out = conv_wino.conv2d([g(s_, cuda) for s_ in srcs], g(wt, cuda), g(b, cuda), act=act, slope=0.1,
                               residual=None if res is None else g(res, cuda), chan_partial=use_part)
    assert list(prof.summary()) == [f"conv3x3_{cin}to{cout}_wino"]
    if use_part:
        out, part = out
        sums = pre.sum(dim=(2, 3))
        assert H.maxabs(part.sum(dim=1).cpu(), sums) <= 2e-6 * sums.abs().max().item() + 2e-3
    assert H.maxabs(out.cpu(), ref) <= 2e-5 * max(1.0, ref.abs().max().item())


def test_conv3x3_winograd_fused_channel_attention_prologue(conv_wino, cuda):
    """conv(r * scale + x) with the side output of the effective input, applied in the Winograd input transform"""
    n, c, h, w = 10, 64, 133, 156
    r, x = cases.randn(1, n, c, h, w), cases.randn(2, n, c, h, w)
    scale = cases.rand(3, n, c)
    wt = cases.randn(4, 64, c, 3, 3, scale=0.05)
    b = cases.randn(5, 64, scale=0.1)
    res = cases.randn(6, n, 64, h, w)
    eff = r * scale.view(n, c, 1, 1) + x
    ref = F.relu(F.conv2d(eff, wt, b, 1, 1))
    assert conv_wino.ca_fusable(torch.zeros(n, c, h, w, device=cuda))
    with conv_wino.profile() as prof:
        out, part, xs = conv_wino.conv2d(g(r, cuda), g(wt, cuda), g(b, cuda), act="relu", ca=(g(scale, cuda), g(x, cuda)),
                                         ca_out=True, chan_partial=True)
    assert list(prof.summary()) == ["conv3x3_64to64_wino_ca"]
    assert H.maxabs(xs.cpu(), eff) <= 1e-6
    assert H.maxabs(out.cpu(), ref) <= 2e-5 * max(1.0, ref.abs().max().item())
    sums = ref.sum(dim=(2, 3))
    assert H.maxabs(part.sum(dim=1).cpu(), sums) <= 2e-6 * sums.abs().max().item() + 2e-3
    out2 = conv_wino.conv2d(g(r, cuda), g(wt, cuda), g(b, cuda), residual=g(res, cuda), ca=(g(scale, cuda), g(x, cuda)))
    assert H.maxabs(out2.cpu(), F.conv2d(eff, wt, b, 1, 1) + res) <= 2e-5 * max(1.0, ref.abs().max().item())


def test_conv3x3_winograd_small_problems_run_the_direct_kernel(conv_wino, cuda):
    x, wt = cases.randn(1, 1, 64, 20, 32), cases.randn(2, 64, 64, 3, 3, scale=0.05)
    with conv_wino.profile() as prof:
        out = conv_wino.conv2d(g(x, cuda), g(wt, cuda), None)
    assert list(prof.summary()) == ["conv3x3_64to64_x6s"]      # (round 5: the small-launch kernel, exact bf16x6)
    assert H.maxabs(out.cpu(), F.conv2d(x, wt, None, 1, 1)) <= 2e-5
    was = conv_wino.CONV3_SMALL
    try:
        conv_wino.CONV3_SMALL = "direct"
        with conv_wino.profile() as prof:
            out = conv_wino.conv2d(g(x, cuda), g(wt, cuda), None)
    finally:
        conv_wino.CONV3_SMALL = was
    assert list(prof.summary()) == ["conv3x3_64to64"]
    assert H.maxabs(out.cpu(), F.conv2d(x, wt, None, 1, 1)) <= 2e-5


def test_conv3x3_winograd_error_against_fp64(ops, cuda):
    """F(2x2, 3x3) rounds differently from the direct sum (transform additions before and after the products); in fp32
    its error stays within a small factor of the direct kernel's."""
    n, h, w = 10, 133, 156
    x = cases.randn(1, n, 64, h, w) * 2.0 + 0.7
    wt = cases.randn(2, 64, 64, 3, 3, scale=0.05)
    ref64 = F.conv2d(x.double(), wt.double(), None, 1, 1)
    scale = ref64.abs().max().item()
    ops.set_conv_mode("direct")
    e_native = (ops.conv2d(g(x, cuda), g(wt, cuda), None).cpu().double() - ref64).abs().max().item() / scale
    ops.set_conv_mode("winograd")
    try:
        e_wino = (ops.conv2d(g(x, cuda), g(wt, cuda), None).cpu().double() - ref64).abs().max().item() / scale
    finally:
        ops.set_conv_mode(DEFAULT_CONV_MODE)
    print("relative max error vs fp64: direct", e_native, "winograd", e_wino)
    assert e_native < 3e-6 and e_wino < 6e-6, (e_native, e_wino)


# ---- 3x3 conv by Winograd F(4x4, 3x3): same descriptor / tensors / epilogue, 8 x 64-pixel tiles -------------------
@pytest.fixture()
def conv_wino4(ops):
    was5 = ops.CONV5_MODE
    ops.set_conv_mode("winograd4")
    ops.CONV5_MODE = "wino"          # (the 5x5 heads default to eavsr_conv_f32x6; these tests are about F(2x2,5x5))
    yield ops
    ops.CONV5_MODE = was5
    ops.set_conv_mode(DEFAULT_CONV_MODE)


@pytest.mark.parametrize("case", [([64], 64, "relu", True, True, 10, 133, 156), ([64, 64], 64, "lrelu", True, False, 10, 133, 156),
                                  ([64], 256, None, False, False, 4, 100, 128), ([64, 64, 64, 64, 64], 64, "lrelu", False, False, 3, 180, 320),
                                  ([8], 40, None, False, True, 13, 65, 68), ([128], 64, "relu", False, True, 1, 400, 320),
                                  ([64], 64, None, True, True, 4, 180, 320), ([4, 12], 7, "relu", False, False, 9, 37, 200)],
                         ids=lambda c: f"c{'+'.join(map(str, c[0]))}_o{c[1]}_{c[5]}x{c[6]}x{c[7]}")
def test_conv3x3_winograd4_vs_torch_cpu(conv_wino4, cuda, case):
    chans, cout, act, use_res, use_part, n, h, w = case
    cin = sum(chans)
    srcs = [cases.randn(10 + i, n, c, h, w) for i, c in enumerate(chans)]
    wt = cases.randn(20, cout, cin, 3, 3, scale=1.0 / (cin * 9) ** 0.5)
    b = cases.randn(21, cout, scale=0.1)
    res = cases.randn(22, n, cout, h, w) if use_res else None
    ref = F.conv2d(torch.cat(srcs, 1), wt, b, 1, 1)
    ref = F.relu(ref) if act == "relu" else (F.leaky_relu(ref, 0.1) if act == "lrelu" else ref)
    pre = ref
    if use_res:
        ref = ref + res
    with conv_wino4.profile() as prof:
        out = conv_wino4.conv2d([g(s_, cuda) for s_ in srcs], g(wt, cuda), g(b, cuda), act=act, slope=0.1,
                                residual=None if res is None else g(res, cuda), chan_partial=use_part)
    assert list(prof.summary()) == [f"conv3x3_{cin}to{cout}_wino4"]
    if use_part:
        out, part = out
        assert part.shape[1] == conv_wino4.lib().eavsr_conv3x3_wino4_tiles(h, w)
        sums = pre.sum(dim=(2, 3))
        assert H.maxabs(part.sum(dim=1).cpu(), sums) <= 1e-5 * sums.abs().max().item() + 5e-3
    assert H.maxabs(out.cpu(), ref) <= 6e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("chans", [[12], [4, 8, 8], [16, 24]])
def test_conv3x3_winograd4_c_abi_odd_and_even_chunk_counts(conv_wino4, cuda, chans):
    """eavsr_conv3x3_wino4_f32 straight through the C ABI with sources of 4-channel granularity: an odd number of chunks runs the
    duty-pair schedule, an even number the grouped one (transform phases of two chunks) -- `ops.conv2d` only ever sends multiples of
    8 channels, so the odd case is reachable from the boundary alone."""
    from eavsr_amd import _native as Nn
    ops = conv_wino4
    n, h, w, cout = 3, 61, 132, 64
    cin = sum(chans)
    srcs = [g(cases.randn(70 + i, n, c, h, w), cuda) for i, c in enumerate(chans)]
    wt = cases.randn(75, cout, cin, 3, 3, scale=1.0 / (cin * 9) ** 0.5)
    b = cases.randn(76, cout, scale=0.1)
    wu = ops._packed_wino([g(wt, cuda)], four=True)
    bg = g(b, cuda)
    out = torch.empty(n, cout, h, w, device=cuda)
    d = Nn.ConvDesc()
    for i, s_ in enumerate(srcs):
        d.src[i] = s_.data_ptr(); d.src_c[i] = chans[i]
    d.n_src = len(chans); d.ksize = 3
    d.bias = bg.data_ptr(); d.out = out.data_ptr()
    d.n, d.h, d.w, d.cin, d.cout = n, h, w, cin, cout
    d.act = 1
    import ctypes as C
    with torch.cuda.device(cuda):
        code = ops.lib().eavsr_conv3x3_wino4_f32(C.byref(d), C.c_void_p(wu.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert code == 0, ops.lib().eavsr_last_error()
    ref = F.relu(F.conv2d(torch.cat([s_.cpu() for s_ in srcs], 1), wt, b, 1, 1))
    assert H.maxabs(out.cpu(), ref) <= 6e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("case", [(64, 256, "lrelu", 4, 100, 128, True), (64, 256, None, 2, 180, 320, True), (64, 40, "relu", 3, 97, 132, True),
                                  (16, 8, "lrelu", 1, 9, 12, False)],
                         ids=lambda c: f"c{c[0]}_o{c[1]}_{c[3]}x{c[4]}x{c[5]}")
def test_conv3x3_pixel_shuffle_epilogue(conv_wino4, cuda, case):
    """the upsampling tail (eavsrp_model.py:343-347: conv -> PixelShuffle(2) -> LeakyReLU): `pixel_shuffle2=True` makes the
    F(4x4,3x3) kernel store F.pixel_shuffle(out, 2) itself -- bit-identical to shuffling the plain output, and equal to the torch
    CPU reference; shapes the kernel does not take fall back to torch's shuffle"""
    cin, cout, act, n, h, w, fused = case
    x = cases.randn(30, n, cin, h, w)
    wt = cases.randn(31, cout, cin, 3, 3, scale=1.0 / (cin * 9) ** 0.5)
    b = cases.randn(32, cout, scale=0.1)
    ref = F.conv2d(x, wt, b, 1, 1)
    ref = F.relu(ref) if act == "relu" else (F.leaky_relu(ref, 0.1) if act == "lrelu" else ref)
    ref = F.pixel_shuffle(ref, 2)
    xg, wg, bg = g(x, cuda), g(wt, cuda), g(b, cuda)
    with conv_wino4.profile() as prof:
        out = conv_wino4.conv2d(xg, wg, bg, act=act, slope=0.1, pixel_shuffle2=True)
    assert (list(prof.summary()) == [f"conv3x3_{cin}to{cout}_wino4"]) == fused
    plain = conv_wino4.conv2d(xg, wg, bg, act=act, slope=0.1)
    assert tuple(out.shape) == (n, cout // 4, 2 * h, 2 * w)
    assert torch.equal(out, F.pixel_shuffle(plain, 2))
    assert H.maxabs(out.cpu(), ref) <= 6e-5 * max(1.0, ref.abs().max().item())
    with pytest.raises(ValueError):
        conv_wino4.conv2d(xg, wg, bg, residual=plain, pixel_shuffle2=True)


@pytest.mark.parametrize("case", [([64], [32, 16, 72], None, False, False, 4, 180, 320), ([64], [64], "lrelu", True, True, 5, 133, 156),
                                  ([4, 12], [7], "relu", False, False, 16, 61, 132), ([64, 64], [130], None, False, True, 6, 90, 160)],
                         ids=lambda c: f"c{'+'.join(map(str, c[0]))}_o{'+'.join(map(str, c[1]))}_{c[5]}x{c[6]}x{c[7]}")
def test_conv5x5_winograd_vs_torch_cpu(conv_wino4, cuda, case):
    """F(2x2, 5x5): the predictor's 5x5 heads (three weights stacked along cout in one launch), ragged sizes, epilogue"""
    chans, couts, act, use_res, use_part, n, h, w = case
    cin, cout = sum(chans), sum(couts)
    srcs = [cases.randn(10 + i, n, c, h, w) for i, c in enumerate(chans)]
    wts = [cases.randn(20 + i, co, cin, 5, 5, scale=1.0 / (cin * 25) ** 0.5) for i, co in enumerate(couts)]
    bs = [cases.randn(30 + i, co, scale=0.1) for i, co in enumerate(couts)]
    res = cases.randn(22, n, cout, h, w) if use_res else None
    ref = F.conv2d(torch.cat(srcs, 1), torch.cat(wts, 0), torch.cat(bs, 0), 1, 2)
    ref = F.relu(ref) if act == "relu" else (F.leaky_relu(ref, 0.1) if act == "lrelu" else ref)
    pre = ref
    if use_res:
        ref = ref + res
    with conv_wino4.profile() as prof:
        out = conv_wino4.conv2d([g(s_, cuda) for s_ in srcs], [g(x, cuda) for x in wts], [g(x, cuda) for x in bs], act=act,
                                slope=0.1, residual=None if res is None else g(res, cuda), chan_partial=use_part)
    assert list(prof.summary()) == [f"conv5x5_{cin}to{cout}_wino"]
    if use_part:
        out, part = out
        assert part.shape[1] == conv_wino4.lib().eavsr_conv5x5_wino_tiles(h, w)
        sums = pre.sum(dim=(2, 3))
        assert H.maxabs(part.sum(dim=1).cpu(), sums) <= 1e-5 * sums.abs().max().item() + 5e-3
    assert H.maxabs(out.cpu(), ref) <= 3e-5 * max(1.0, ref.abs().max().item())
    # error against fp64 for the path's head shape: same order as the direct kernel's
    if cout == 120:
        ref64 = F.conv2d(torch.cat(srcs, 1).double(), torch.cat(wts, 0).double(), torch.cat(bs, 0).double(), 1, 2)
        e5 = (out.cpu().double() - ref64).abs().max().item() / ref64.abs().max().item()
        print("5x5 winograd relative max error vs fp64:", e5)
        assert e5 < 1.5e-5, e5


def test_conv3x3_winograd4_fused_channel_attention_prologue(conv_wino4, cuda):
    """conv(r * scale + x) with the side output of the effective input, applied in the F(4x4, 3x3) input transform;
    ragged size (tiles cut by the image edge), several tiles per workgroup, with and without side output / residual"""
    conv_wino4.require_lab("the channel-attention prologue inside the F(4x4,3x3) kernel")
    n, c, h, w = 10, 64, 133, 156
    r, x = cases.randn(1, n, c, h, w), cases.randn(2, n, c, h, w)
    scale = cases.rand(3, n, c)
    wt = cases.randn(4, 64, c, 3, 3, scale=0.05)
    b = cases.randn(5, 64, scale=0.1)
    res = cases.randn(6, n, 64, h, w)
    eff = r * scale.view(n, c, 1, 1) + x
    ref = F.relu(F.conv2d(eff, wt, b, 1, 1))
    with conv_wino4.profile() as prof:
        out, part, xs = conv_wino4.conv2d(g(r, cuda), g(wt, cuda), g(b, cuda), act="relu", ca=(g(scale, cuda), g(x, cuda)),
                                          ca_out=True, chan_partial=True)
    assert list(prof.summary()) == ["conv3x3_64to64_wino4_ca"]
    assert H.maxabs(xs.cpu(), eff) <= 1e-6
    assert H.maxabs(out.cpu(), ref) <= 6e-5 * max(1.0, ref.abs().max().item())
    sums = ref.sum(dim=(2, 3))
    assert H.maxabs(part.sum(dim=1).cpu(), sums) <= 1e-5 * sums.abs().max().item() + 5e-3
    out2 = conv_wino4.conv2d(g(r, cuda), g(wt, cuda), g(b, cuda), residual=g(res, cuda), ca=(g(scale, cuda), g(x, cuda)))
    assert H.maxabs(out2.cpu(), F.conv2d(eff, wt, b, 1, 1) + res) <= 6e-5 * max(1.0, ref.abs().max().item())
    # 128 -> 40 channels, wide image
    r2, x2, s2 = cases.randn(7, 3, 128, 64, 320), cases.randn(8, 3, 128, 64, 320), cases.rand(9, 3, 128)
    w2 = cases.randn(10, 40, 128, 3, 3, scale=0.04)
    o2, xs2 = conv_wino4.conv2d(g(r2, cuda), g(w2, cuda), None, ca=(g(s2, cuda), g(x2, cuda)), ca_out=True)
    eff2 = r2 * s2.view(3, 128, 1, 1) + x2
    assert H.maxabs(xs2.cpu(), eff2) <= 1e-6
    ref2 = F.conv2d(eff2, w2, None, 1, 1)
    assert H.maxabs(o2.cpu(), ref2) <= 6e-5 * max(1.0, ref2.abs().max().item())


def test_conv3x3_winograd4_error_against_fp64_and_fallbacks(ops, cuda):
    """F(4x4, 3x3) in fp32: ~1e-5 of the output scale (the 6 x 6 transforms amplify rounding), the documented price of
    4x fewer multiplications; small problems fall back to the direct kernel."""
    n, h, w = 10, 133, 156
    x = cases.randn(1, n, 64, h, w) * 2.0 + 0.7
    wt = cases.randn(2, 64, 64, 3, 3, scale=0.05)
    ref64 = F.conv2d(x.double(), wt.double(), None, 1, 1)
    scale = ref64.abs().max().item()
    ops.set_conv_mode("winograd4")
    try:
        e4 = (ops.conv2d(g(x, cuda), g(wt, cuda), None).cpu().double() - ref64).abs().max().item() / scale
        print("relative max error vs fp64: winograd4", e4)
        assert e4 < 3e-5, e4
        with ops.profile() as prof:
            xs, wsm = cases.randn(3, 1, 64, 20, 32), cases.randn(4, 64, 64, 3, 3, scale=0.05)
            small = ops.conv2d(g(xs, cuda), g(wsm, cuda), None)
        assert list(prof.summary()) == ["conv3x3_64to64_x6s"]
        assert H.maxabs(small.cpu(), F.conv2d(xs, wsm, None, 1, 1)) <= 2e-5
        r, xx, sc = cases.randn(5, n, 64, h, w), cases.randn(6, n, 64, h, w), cases.rand(7, n, 64)
        with ops.profile() as prof:
            out = ops.conv2d(g(r, cuda), g(wt, cuda), None, ca=(g(sc, cuda), g(xx, cuda)))
        # the prologue inside the Winograd kernel is a lab instantiation; the default build folds it into the direct kernel
        assert list(prof.summary()) == ["conv3x3_64to64_wino4_ca" if ops.lab_available() else "conv3x3_64to64_ca"]
        ref = F.conv2d(r * sc.view(n, 64, 1, 1) + xx, wt, None, 1, 1)
        assert H.maxabs(out.cpu(), ref) <= 6e-5 * max(1.0, ref.abs().max().item())
    finally:
        ops.set_conv_mode(DEFAULT_CONV_MODE)


# ---- 3x3 conv, bf16x9 contraction (opt-in): same descriptor / tensors / epilogue as the native kernel -------------
@pytest.fixture()
def conv_x9(ops):
    ops.set_conv_mode("bf16x9")
    yield ops
    ops.set_conv_mode(DEFAULT_CONV_MODE)


@pytest.mark.parametrize("case", [([64], 64, "relu", True, True), ([64, 64], 64, "lrelu", True, False),
                                  ([64], 256, None, False, False), ([64, 64, 64, 64, 64], 64, "lrelu", False, False),
                                  ([8], 40, None, False, True)], ids=lambda c: f"c{'+'.join(map(str, c[0]))}_o{c[1]}")
def test_conv3x3_x9_vs_torch_cpu(conv_x9, cuda, case):
    chans, cout, act, use_res, use_part = case
    n, h, w = 10, 133, 156                       # 250 tiles of 32 rows: the size class the bf16x9 kernel exists for
    cin = sum(chans)
    srcs = [cases.randn(10 + i, n, c, h, w) for i, c in enumerate(chans)]
    wt = cases.randn(20, cout, cin, 3, 3, scale=1.0 / (cin * 9) ** 0.5)
    b = cases.randn(21, cout, scale=0.1)
    res = cases.randn(22, n, cout, h, w) if use_res else None
    ref = F.conv2d(torch.cat(srcs, 1), wt, b, 1, 1)
    ref = F.relu(ref) if act == "relu" else (F.leaky_relu(ref, 0.1) if act == "lrelu" else ref)
    pre = ref
    if use_res:
        ref = ref + res
    with conv_x9.profile() as prof:
        out = conv_x9.conv2d([g(s_, cuda) for s_ in srcs], g(wt, cuda), g(b, cuda), act=act, slope=0.1,
                             residual=None if res is None else g(res, cuda), chan_partial=use_part)
    assert list(prof.summary()) == [f"conv3x3_{cin}to{cout}_x9"]
    if use_part:
        out, part = out
        sums = pre.sum(dim=(2, 3))
        assert H.maxabs(part.sum(dim=1).cpu(), sums) <= 2e-6 * sums.abs().max().item() + 2e-3
    assert H.maxabs(out.cpu(), ref) <= 2e-5 * max(1.0, ref.abs().max().item())


def test_conv3x3_x9_small_or_unaligned_problems_run_the_native_kernel(conv_x9, cuda):
    x, wt = cases.randn(1, 1, 64, 20, 30), cases.randn(2, 64, 64, 3, 3, scale=0.05)
    with conv_x9.profile() as prof:
        out = conv_x9.conv2d(g(x, cuda), g(wt, cuda), None)
    assert list(prof.summary()) == ["conv3x3_64to64"]
    assert H.maxabs(out.cpu(), F.conv2d(x, wt, None, 1, 1)) <= 2e-5


def test_conv3x3_x9_error_against_fp64_is_that_of_the_fp32_kernel(ops, cuda):
    n, h, w = 10, 133, 156
    x = cases.randn(1, n, 64, h, w) * 2.0 + 0.7
    wt = cases.randn(2, 64, 64, 3, 3, scale=0.05)
    ref64 = F.conv2d(x.double(), wt.double(), None, 1, 1)
    scale = ref64.abs().max().item()
    ops.set_conv_mode("direct")
    e_native = (ops.conv2d(g(x, cuda), g(wt, cuda), None).cpu().double() - ref64).abs().max().item() / scale
    ops.set_conv_mode("bf16x9")
    try:
        e_x9 = (ops.conv2d(g(x, cuda), g(wt, cuda), None).cpu().double() - ref64).abs().max().item() / scale
    finally:
        ops.set_conv_mode(DEFAULT_CONV_MODE)
    assert e_native < 3e-6 and e_x9 < 3e-6, (e_native, e_x9)       # measured: 1.3e-6 native, 0.94e-6 bf16x9
    assert e_x9 <= 1.5 * e_native + 2e-8, (e_native, e_x9)


# ---- a7, bf16x9 contraction (opt-in): exact three-way bf16 split of both operands, nine partial products --------
@pytest.fixture()
def dcn_x9(ops):
    ops.set_dcn_mode("bf16x9")
    yield ops
    ops.set_dcn_mode(DEFAULT_DCN_MODE)


@pytest.mark.parametrize("sigma", [0.0, 0.5, 2.0, 8.0])
@pytest.mark.parametrize("shape", [(1, 64, 24, 40, 64, 8), (2, 64, 13, 36, 64, 8), (1, 64, 10, 12, 64, 1),
                                   (1, 16, 9, 32, 32, 2), (1, 64, 7, 8, 40, 8), (1, 64, 21, 68, 96, 4)])
def test_dcnv2_x9_vs_oracle(dcn_x9, cuda, shape, sigma):
    n, c, h, w, cout, dg = shape
    x, off, mask, wt, b = _dcn_inputs(n, c, h, w, cout, dg, sigma)
    ref = O.dcnv2(x, off, mask, wt, b, 1, 1, 1, 1, dg)
    with dcn_x9.profile() as prof:
        out = dcn_x9.modulated_deform_conv2d(g(x, cuda), g(off, cuda), g(mask, cuda), g(wt, cuda), g(b, cuda), 1, 1, 1, 1, dg)
    assert list(prof.summary()) == ["dcnv2_x9"]                # the bf16x9 kernel ran, not the native one
    assert H.maxabs(out.cpu(), ref) <= 3e-5 * max(1.0, ref.abs().max().item())


def test_dcnv2_x9_falls_back_to_native_kernel_on_unaligned_width(dcn_x9, cuda):
    x, off, mask, wt, b = _dcn_inputs(1, 64, 9, 37, 64, 8, 1.0)   # w % 4 != 0: no LDS window, native kernel
    ref = O.dcnv2(x, off, mask, wt, b, 1, 1, 1, 1, 8)
    with dcn_x9.profile() as prof:
        out = dcn_x9.modulated_deform_conv2d(g(x, cuda), g(off, cuda), g(mask, cuda), g(wt, cuda), g(b, cuda), 1, 1, 1, 1, 8)
    assert list(prof.summary()) == ["dcnv2"]
    assert H.maxabs(out.cpu(), ref) <= 3e-5 * max(1.0, ref.abs().max().item())


def test_dcnv2_x9_error_against_fp64_is_that_of_the_fp32_kernel(ops, cuda):
    """The split is exact and all nine partial products are kept, so against an fp64 evaluation of the same sums
    the bf16x9 kernel must be as accurate as the native fp32-MFMA kernel (both only round in the accumulation) --
    and far inside what rounding the operands to bf16 would give (~4e-3 relative)."""
    x, off, mask, wt, b = _dcn_inputs(2, 64, 32, 64, 64, 8, 1.5, seed=10)
    x = x * 3.0 + 0.5                                          # non-zero mean: the running sums are large
    ref64 = O.dcnv2(x.double(), off.double(), mask.double(), wt.double(), b.double(), 1, 1, 1, 1, 8)
    scale = ref64.abs().max().item()
    args = (g(x, cuda), g(off, cuda), g(mask, cuda), g(wt, cuda), g(b, cuda), 1, 1, 1, 1, 8)
    ops.set_dcn_mode("native")
    e_native = (ops.modulated_deform_conv2d(*args).cpu().double() - ref64).abs().max().item() / scale
    ops.set_dcn_mode("bf16x9")
    try:
        e_x9 = (ops.modulated_deform_conv2d(*args).cpu().double() - ref64).abs().max().item() / scale
    finally:
        ops.set_dcn_mode(DEFAULT_DCN_MODE)
    assert e_native < 5e-6 and e_x9 < 5e-6, (e_native, e_x9)      # measured: 2.1e-6 and 2.2e-6 (sampler rounding dominates)
    assert e_x9 <= 1.5 * e_native + 1e-7, (e_native, e_x9)


@pytest.mark.parametrize("preset", ["default", "trained_like"])
@pytest.mark.parametrize("shape", [(1, 64, 12, 16), (2, 64, 45, 80), (1, 64, 33, 70), (1, 64, 5, 3), (1, 8, 20, 40)])
def test_flow_level_fused_kernel_vs_oracle(ops, cuda, shape, preset):
    """TransOffsetworelu(AdaptBlock2_3x3(x, h_hr)) (networks.py:334-348, 566-571) as one kernel: tile interiors, image
    borders (every stage zero-pads its own input), ragged sizes, sizes below one tile"""
    n, c, h, w = shape
    shapes = {"f.regular_matrix": (2, 9), "f.concat.0.weight": (2 * c, 1, 3, 3), "f.concat.0.bias": (2 * c,),
              "f.concat2.0.weight": (c, 2, 3, 3), "f.concat2.0.bias": (c,),
              "f.transform_matrix_conv.weight": (4, c, 3, 3), "f.transform_matrix_conv.bias": (4,),
              "f.translation_conv.weight": (2, c, 3, 3), "f.translation_conv.bias": (2,),
              "t.conv_first.weight": (2, 18, 3, 3), "t.conv_first.bias": (2,)}
    sd = H.filled(shapes, preset)
    x, hh = cases.randn(1, n, c, h, w), cases.randn(2, n, c, h, w)
    ref = O.trans_offset(sd, "t.", O.adapt_block2_3x3(sd, "f.", x, hh))
    d = lambda k: g(sd[k], cuda)
    out = ops.flow_level(g(x, cuda), g(hh, cuda), d("f.concat.0.weight"), d("f.concat.0.bias"), d("f.concat2.0.weight"),
                         d("f.concat2.0.bias"), [d("f.transform_matrix_conv.weight"), d("f.translation_conv.weight")],
                         [d("f.transform_matrix_conv.bias"), d("f.translation_conv.bias")], d("t.conv_first.weight"),
                         d("t.conv_first.bias")).cpu()
    assert H.maxabs(out, ref) <= 3e-5 * max(1.0, ref.abs().max().item())


def test_flow_level_golden(ops, cuda):
    """the reference's own AdaptBlock2_3x3 -> TransOffsetworelu output (golden G2 `flow2`) through the fused kernel"""
    for preset in ("default", "trained_like"):
        g2 = H.golden(f"g2_adapt3x3_{preset}")
        sd = H.filled({**H.adapt3x3_shapes("g2.flow."), **H.trans_shapes("g2.trans.")}, preset)
        x, hh = cases.g2_inputs()
        d = lambda k: g(sd[k], cuda)
        out = ops.flow_level(g(x, cuda), g(hh, cuda), d("g2.flow.concat.0.weight"), d("g2.flow.concat.0.bias"),
                             d("g2.flow.concat2.0.weight"), d("g2.flow.concat2.0.bias"),
                             [d("g2.flow.transform_matrix_conv.weight"), d("g2.flow.translation_conv.weight")],
                             [d("g2.flow.transform_matrix_conv.bias"), d("g2.flow.translation_conv.bias")],
                             d("g2.trans.conv_first.weight"), d("g2.trans.conv_first.bias")).cpu()
        assert H.maxabs(out, g2["flow2"]) <= 2e-5, preset


# ---- round-2 hot-path DCNv2: IL8 input layout, bf16 x6 / x9 products, optional fused affine + sigmoid ("heads") ----------
def test_to_il8_layout(ops, cuda):
    x = cases.randn(3, 2, 24, 7, 9)
    il = ops.to_il8(g(x, cuda)).cpu()
    assert il.shape == (2, 3, 7, 9, 8)
    assert torch.equal(il, x.view(2, 3, 8, 7, 9).permute(0, 1, 3, 4, 2))


@pytest.fixture(params=["il2", "il", "ws"])
def il_impl(request, ops):
    """every schedule of the IL8 DCNv2 kernel: eavsr_dcnv2_il2_f32 (round 4, default), eavsr_dcnv2_il_f32 (round 2) and the
    wave-specialised eavsr_dcnv2_ws_f32"""
    prev = ops.DCN_IL_IMPL
    ops.set_dcn_il_impl(request.param)
    yield request.param
    ops.set_dcn_il_impl(prev)


@pytest.mark.parametrize("nprod", [6, 9])
@pytest.mark.parametrize("sigma", [0.0, 0.5, 2.0, 8.0])
@pytest.mark.parametrize("shape", [(1, 64, 24, 40, 64, 8), (2, 64, 13, 37, 64, 8), (1, 64, 10, 12, 64, 1),
                                   (1, 16, 9, 33, 32, 2), (1, 64, 7, 5, 40, 8), (1, 64, 21, 68, 96, 4),
                                   (3, 64, 45, 80, 64, 8)])
def test_dcnv2_il_vs_oracle(ops, cuda, shape, sigma, nprod, il_impl):
    """explicit offsets / mask (mmcv's signature) through the IL8 kernel: any width (no w % 4 restriction), several tiles per
    persistent workgroup, ragged edges, out-of-window taps (sigma = 8) through the global fix-up"""
    n, c, h, w, cout, dg = shape
    x, off, mask, wt, b = _dcn_inputs(n, c, h, w, cout, dg, sigma)
    ref = O.dcnv2(x, off, mask, wt, b, 1, 1, 1, 1, dg)
    out = ops.dcnv2_il(ops.to_il8(g(x, cuda)), g(off, cuda), g(mask, cuda), g(wt, cuda), g(b, cuda), dg, nprod=nprod)
    assert H.maxabs(out.cpu(), ref) <= 3e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("nprod", [6, 9])
@pytest.mark.parametrize("shape", [(1, 24, 40, 8), (2, 13, 37, 8), (1, 45, 80, 8), (1, 9, 11, 2)])
def test_dcnv2_il_heads_mode_applies_the_affine_expansion_and_the_sigmoid(ops, cuda, shape, nprod, il_impl):
    """heads mode == AdaptBlockOffset's tail (networks.py:302-315: offset = T.R - R + t per group, mask = sigmoid) followed
    by DCNv2, without de_offset / mask ever existing in memory"""
    n, h, w, D = shape
    c = 8 * D
    x = cases.randn(11, n, c, h, w)
    heads = torch.cat([cases.randn(12, n, 4 * D, h, w, scale=0.4) + torch.tensor([1.0, 0, 0, 1.0]).repeat(D).view(1, 4 * D, 1, 1),
                       cases.randn(13, n, 2 * D, h, w, scale=1.5), cases.randn(14, n, 9 * D, h, w, scale=2.0)], 1)
    wt = cases.randn(15, 64, c, 3, 3, scale=1.0 / (c * 9) ** 0.5)
    b = cases.randn(16, 64, scale=0.1)
    off = O.affine_offsets(heads[:, :4 * D], heads[:, 4 * D:6 * D], D)
    mask = torch.sigmoid(heads[:, 6 * D:])
    ref = O.dcnv2(x, off, mask, wt, b, 1, 1, 1, 1, D)
    out = ops.dcnv2_il(ops.to_il8(g(x, cuda)), g(heads, cuda), None, g(wt, cuda), g(b, cuda), D, nprod=nprod, heads=True)
    assert H.maxabs(out.cpu(), ref) <= 3e-5 * max(1.0, ref.abs().max().item())
    # and the explicit form of the same thing agrees with it to rounding
    out2 = ops.dcnv2_il(ops.to_il8(g(x, cuda)), g(off, cuda), g(mask, cuda), g(wt, cuda), g(b, cuda), D, nprod=nprod)
    assert H.maxabs(out.cpu(), out2.cpu()) <= 2e-5 * max(1.0, ref.abs().max().item())


def test_dcnv2_il2_takes_masks_activated_by_the_heads_convolution(ops, cuda):
    """heads = 2 (round 4): the 5x5 heads convolution applies the mask sigmoid in its epilogue (networks.py:313-314, `sigmoid_from`)
    and eavsr_dcnv2_il2_f32 takes the masks as they are -- the same instructions evaluate the sigmoid on either side, so the
    outputs are bit-identical to heads = 1 on the logits"""
    n, h, w, D = 2, 45, 80, 8
    c = 8 * D
    x = cases.randn(21, n, c, h, w)
    f = cases.randn(22, n, 64, h, w, scale=0.5)
    ws = [cases.randn(23, 4 * D, 64, 5, 5, scale=0.01), cases.randn(24, 2 * D, 64, 5, 5, scale=0.02), cases.randn(25, 9 * D, 64, 5, 5, scale=0.03)]
    bs = [torch.tensor([1.0, 0, 0, 1.0]).repeat(D), cases.randn(26, 2 * D, scale=0.5), cases.randn(27, 9 * D, scale=0.5)]
    wt = cases.randn(28, 64, c, 3, 3, scale=1.0 / (c * 9) ** 0.5)
    b = cases.randn(29, 64, scale=0.1)
    gw, gb = [g(t, cuda) for t in ws], [g(t, cuda) for t in bs]
    prev = ops.DCN_IL_IMPL
    ops.set_dcn_il_impl("il2")
    try:
        logits = ops.conv2d(g(f, cuda), gw, gb)
        masks = ops.conv2d(g(f, cuda), gw, gb, sigmoid_from=6 * D)
        assert torch.equal(logits[:, :6 * D], masks[:, :6 * D])
        ref_m = torch.sigmoid(logits[:, 6 * D:].double()).float()
        assert H.maxabs(masks[:, 6 * D:].cpu(), ref_m.cpu()) <= 3e-7
        xil = ops.to_il8(g(x, cuda))
        o1 = ops.dcnv2_il(xil, logits, None, g(wt, cuda), g(b, cuda), D, heads=True)
        o2 = ops.dcnv2_il(xil, masks, None, g(wt, cuda), g(b, cuda), D, heads=True, mask_activated=True)
        assert torch.equal(o1, o2)
        # against the oracle through the explicit form
        off = O.affine_offsets(logits[:, :4 * D].cpu(), logits[:, 4 * D:6 * D].cpu(), D)
        ref = O.dcnv2(x, off, torch.sigmoid(logits[:, 6 * D:].cpu()), wt, b, 1, 1, 1, 1, D)
        assert H.maxabs(o2.cpu(), ref) <= 3e-5 * max(1.0, ref.abs().max().item())
        ops.set_dcn_il_impl("il")
        with pytest.raises(ValueError):
            ops.dcnv2_il(xil, masks, None, g(wt, cuda), g(b, cuda), D, heads=True, mask_activated=True)
    finally:
        ops.set_dcn_il_impl(prev)
    # every other route of conv2d applies the same sigmoid by torch
    w3 = cases.randn(30, 16, 64, 3, 3, scale=0.05)
    y3 = ops.conv2d(g(f, cuda), g(w3, cuda), None, sigmoid_from=8).cpu()
    r3 = F.conv2d(f, w3, None, 1, 1)
    r3[:, 8:] = torch.sigmoid(r3[:, 8:])
    assert H.maxabs(y3, r3) <= 1e-4


def test_dcnv2_il_error_against_fp64(ops, cuda):
    """x9 keeps every partial product (exact operands, fp32 accumulation); x6 drops the three products below 2^-23 of the
    result.  Against an fp64 evaluation both must be as accurate as the native fp32-MFMA kernel."""
    x, off, mask, wt, b = _dcn_inputs(2, 64, 32, 64, 64, 8, 1.5, seed=10)
    x = x * 3.0 + 0.5
    ref64 = O.dcnv2(x.double(), off.double(), mask.double(), wt.double(), b.double(), 1, 1, 1, 1, 8)
    scale = ref64.abs().max().item()
    args = (g(x, cuda), g(off, cuda), g(mask, cuda), g(wt, cuda), g(b, cuda), 1, 1, 1, 1, 8)
    errs = {}
    prev = ops.DCN_MODE
    try:
        for mode in ("native", "il9", "il6"):
            ops.set_dcn_mode(mode)
            errs[mode] = (ops.modulated_deform_conv2d(*args).cpu().double() - ref64).abs().max().item() / scale
    finally:
        ops.set_dcn_mode(prev)
    print("DCNv2 relative max error vs fp64:", errs)
    assert all(e < 5e-6 for e in errs.values()), errs
    assert errs["il9"] <= 1.5 * errs["native"] + 1e-7 and errs["il6"] <= 1.5 * errs["native"] + 2e-7, errs


def test_dcnv2_il_bad_arguments(ops, cuda):
    z = lambda *s_: torch.zeros(*s_, device=cuda)
    with pytest.raises(ValueError):
        ops.dcnv2_il(z(1, 8, 8, 8, 4), z(1, 144, 8, 8), z(1, 72, 8, 8), z(64, 64, 3, 3), None, 8)       # not IL8
    with pytest.raises(ValueError):
        ops.dcnv2_il(z(1, 8, 8, 8, 8), z(1, 100, 8, 8), None, z(64, 64, 3, 3), None, 8, heads=True)     # heads channels
    with pytest.raises(RuntimeError):
        ops.dcnv2_il(z(1, 8, 8, 8, 8), z(1, 144, 8, 8), z(1, 72, 8, 8), z(64, 64, 3, 3), None, 8, nprod=7)


# ------------------------------------------------------------------------------------------ a3/a6 pieces
@pytest.mark.parametrize("shape", [(1, 64, 12, 16), (2, 64, 45, 80), (1, 64, 33, 130)])
def test_adapt_frontend_vs_oracle(ops, cuda, shape):
    n, c, h, w = shape
    sd = H.filled(H.adapt_front_shapes("f."), "trained_like")
    x, hh = cases.randn(1, n, c, h, w), cases.randn(2, n, c, h, w)
    ref = O.adapt_frontend(sd, "f.", x, hh)
    out = ops.adapt_frontend(g(x, cuda), g(hh, cuda), g(sd["f.concat.0.weight"], cuda), g(sd["f.concat.0.bias"], cuda),
                             g(sd["f.concat2.0.weight"], cuda), g(sd["f.concat2.0.bias"], cuda)).cpu()
    assert H.maxabs(out, ref) <= 1e-5


@pytest.mark.parametrize("D,with_mask", [(1, False), (8, True)])
def test_affine_offsets_vs_oracle(ops, cuda, D, with_mask):
    n, h, w = 2, 11, 19
    heads = cases.randn(1, n, (15 if with_mask else 6) * D, h, w)
    off, mask = ops.affine_offsets(g(heads, cuda), D, with_mask)
    ref = O.affine_offsets(heads[:, :4 * D], heads[:, 4 * D:6 * D], D)
    assert H.maxabs(off.cpu(), ref) <= 1e-5
    if with_mask:
        assert H.maxabs(mask.cpu(), torch.sigmoid(heads[:, 6 * D:])) <= 1e-6


# ------------------------------------------------------------------------------------------ a5/a12 glue
@pytest.mark.parametrize("hin,win,hout,wout,scale", [(32, 48, 8, 12, 0.25), (32, 48, 16, 24, 0.5), (8, 12, 16, 24, 2.0),
                                                      (45, 80, 90, 160, 2.0), (6, 10, 12, 20, 2.0)])
def test_resize_bilinear_ac(ops, cuda, hin, win, hout, wout, scale):
    x = cases.randn(1, 2, 2, hin, win)
    pre = cases.randn(2, 2, 2, hin, win)
    post = cases.randn(3, 2, 2, hout, wout)
    ref = F.interpolate(x + pre, size=(hout, wout), mode="bilinear", align_corners=True) * scale + post
    out = ops.resize_bilinear_ac(g(x, cuda), (hout, wout), scale, pre_add=g(pre, cuda), post_add=g(post, cuda)).cpu()
    assert H.maxabs(out, ref) <= 1e-5
    ref2 = F.interpolate(x, size=(hout, wout), mode="bilinear", align_corners=True) * scale
    assert H.maxabs(ops.resize_bilinear_ac(g(x, cuda), (hout, wout), scale).cpu(), ref2) <= 1e-5


def test_pyramid_matches_interpolate(ops, cuda):
    x = cases.randn(1, 3, 64, 36, 52)
    d2, d4 = ops.pyramid(g(x, cuda))
    r2, r4 = O.feature_pyramid(x)
    assert H.maxabs(d2.cpu(), r2) <= 1e-6 and H.maxabs(d4.cpu(), r4) <= 1e-6
    with pytest.raises(ValueError):
        ops.pyramid(torch.zeros(1, 2, 10, 12, device=cuda))


def test_glue_either_side_of_the_path_matches_aten(ops, cuda):
    """round 3: the ATen kernels that were left inside the forward (SPyNet's normalisation, 2x2 average-pool pyramid, resize to a
    multiple of 32 and back, 8-channel concat; the encoder's normalisation; the tail's bilinear LR skip) as HIP kernels,
    against the torch CPU ops the reference calls (models/eavsrp_model.py:436-437,450-462,486,499-521,158,359)."""
    x = cases.randn(5, 4, 3, 60, 100) * 0.3 + 0.5
    mean, std = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    assert H.maxabs(ops.normalize(g(x, cuda), g(mean, cuda), g(std, cuda)).cpu(), (x - mean) / std) <= 1e-6
    assert H.maxabs(ops.avg_pool2(g(x, cuda)).cpu(), F.avg_pool2d(x, 2, 2, count_include_pad=False)) <= 1e-6
    with pytest.raises(ValueError):
        ops.avg_pool2(torch.zeros(1, 1, 5, 4, device=cuda))
    for size in ((64, 128), (60, 100), (240, 400), (37, 51)):       # up to /32 multiples, identity, x4, an odd shrink
        got = ops.resize_bilinear(g(x, cuda), size).cpu()
        want = F.interpolate(x, size=size, mode="bilinear", align_corners=False)
        assert H.maxabs(got, want) <= 2e-6, size
    want4 = torch.nn.Upsample(scale_factor=4, mode="bilinear", align_corners=False)(x)
    assert H.maxabs(ops.resize_bilinear(g(x, cuda), (240, 400)).cpu(), want4) <= 2e-6
    fl = cases.randn(6, 3, 2, 64, 128) * 3
    got = ops.resize_bilinear(g(fl, cuda), (60, 100), channel_mul=(100 / 128, 60 / 64)).cpu()
    want = F.interpolate(fl, size=(60, 100), mode="bilinear", align_corners=False)
    want[:, 0] *= float(100) / float(128)
    want[:, 1] *= float(60) / float(64)
    assert H.maxabs(got, want) <= 2e-6
    a, b, c = cases.randn(7, 3, 3, 12, 20), cases.randn(8, 3, 3, 12, 20), cases.randn(9, 3, 2, 12, 20)
    assert torch.equal(ops.concat3(g(a, cuda), g(b, cuda), g(c, cuda)).cpu(), torch.cat([a, b, c], 1))


def test_add(ops, cuda):
    a, b, c = (cases.randn(i, 3, 2, 17, 19) for i in range(3))
    assert torch.equal(ops.add(g(a, cuda), g(b, cuda)).cpu(), a + b)
    assert torch.equal(ops.add(g(a, cuda), g(b, cuda), g(c, cuda)).cpu(), a + b + c)


# ------------------------------------------------------------------------------------------ a11
def test_channel_attention_pieces(ops, cuda):
    sd = H.filled(H.rcab_shapes("b."), "trained_like")
    x = cases.randn(1, 2, 64, 20, 28)
    r = cases.randn(2, 2, 64, 20, 28)
    part = r.view(2, 64, 4, -1).sum(-1).permute(0, 2, 1).contiguous()  # 4 fake tiles
    scale = ops.ca_scale(g(part, cuda), 20 * 28, g(sd["b.ca.conv_du.0.weight"], cuda), g(sd["b.ca.conv_du.0.bias"], cuda),
                         g(sd["b.ca.conv_du.2.weight"], cuda), g(sd["b.ca.conv_du.2.bias"], cuda))
    ref = O.ca_layer(sd, "b.ca.", r) + x
    out = ops.scale_residual(g(r, cuda), scale, g(x, cuda)).cpu()
    assert H.maxabs(out, ref) <= 1e-5


def test_conv2d_fused_channel_attention_prologue(ops, cuda):
    """conv(r * scale + x) with the side output of the effective input == scale_residual followed by conv
    (the fused prologue exists for the 32-row tile, which the host picks when it fills the CUs: 10 x 5 x 5 tiles here)"""
    n, c, h, w = 10, 64, 133, 156
    assert ops.ca_fusable(torch.zeros(n, c, h, w, device=cuda))
    r, x = cases.randn(1, n, c, h, w), cases.randn(2, n, c, h, w)
    scale = cases.rand(3, n, c)
    wt = cases.randn(4, 64, c, 3, 3, scale=0.05)
    b = cases.randn(5, 64, scale=0.1)
    res = cases.randn(6, n, 64, h, w)
    eff = r * scale.view(n, c, 1, 1) + x
    ref = F.relu(F.conv2d(eff, wt, b, 1, 1))
    out, xs = ops.conv2d(g(r, cuda), g(wt, cuda), g(b, cuda), act="relu", ca=(g(scale, cuda), g(x, cuda)), ca_out=True)
    assert H.maxabs(xs.cpu(), eff) <= 1e-6
    assert H.maxabs(out.cpu(), ref) <= 2e-5 * max(1.0, ref.abs().max().item())
    out2 = ops.conv2d(g(r, cuda), g(wt, cuda), g(b, cuda), residual=g(res, cuda), ca=(g(scale, cuda), g(x, cuda)))
    assert H.maxabs(out2.cpu(), F.conv2d(eff, wt, b, 1, 1) + res) <= 2e-5 * max(1.0, ref.abs().max().item())
    assert not ops.ca_fusable(torch.zeros(1, 64, 8, 10, device=cuda))
    with pytest.raises(NotImplementedError):
        ops.conv2d(torch.zeros(1, 64, 8, 10, device=cuda), g(wt, cuda), None,
                   ca=(torch.zeros(1, 64, device=cuda), torch.zeros(1, 64, 8, 10, device=cuda)))


@pytest.mark.parametrize("shape", [(2, 64, 180, 320), (1, 64, 24, 40), (3, 64, 45, 80), (1, 32, 12, 16), (2, 64, 9, 13), (1, 64, 136, 240)])
def test_ca_tail_one_launch_equals_ca_scale_plus_scale_residual(ops, cuda, shape):
    """eavsr_ca_tail_f32 (CALayer + `res * y + x`, networks.py:444-447,463-464, in one launch) against the two-launch form
    bit for bit (same fixed-order reduction, same fma) and against the CPU oracle's ca_layer arithmetic"""
    n, c, h, w = shape
    r, x = cases.randn(70, n, c, h, w), cases.randn(71, n, c, h, w)
    cr = max(c // 16, 1)
    w1, b1 = cases.randn(72, cr, c, 1, 1, scale=0.2), cases.randn(73, cr, scale=0.1)
    w2, b2 = cases.randn(74, c, cr, 1, 1, scale=0.5), cases.randn(75, c, scale=0.1)
    # 115: the bench shape's tile count (both short loops of the fixed-order reduction); 1020: configs[4]'s (the sixteen-deep loop too)
    tiles = 115 if h == 180 else 67 if h == 45 else 1020 if h == 136 else 7
    # per-tile channel sums that add up to the true sums (what the conv epilogue hands over)
    sums = r.sum(dim=(2, 3))
    frac = torch.softmax(cases.randn(76, n, tiles, c), dim=1)
    partial = frac * sums.view(n, 1, c)
    rg, xg, pg = g(r, cuda), g(x, cuda), g(partial, cuda)
    args = [g(t, cuda) for t in (w1, b1, w2, b2)]
    with ops.profile() as prof:
        out = ops.ca_tail(rg, pg, *args, xg)
    fused = "ca_tail" in prof.summary()
    assert fused == ((h * w) % 4 == 0)
    two = ops.scale_residual(rg, ops.ca_scale(pg, h * w, *args), xg)
    assert torch.equal(out, two)
    mean = sums / (h * w)
    y = torch.sigmoid(F.conv2d(F.relu(F.conv2d(mean.view(n, c, 1, 1), w1, b1)), w2, b2))
    ref = r * y + x
    assert H.maxabs(out.cpu(), ref) <= 2e-5 * max(1.0, ref.abs().max().item())
