"""GPU parity tests of the backward kernels (training step): gradients from the
HIP backward kernels (through eavsr_amd.autograd) against CPU autograd through the oracle's restatement of
the same functions.  `pytest -m gpu`."""
from argparse import Namespace

import pytest
import torch
import torch.nn.functional as F

from oracle import eavsr_oracle as O
from tests import helpers as H
from tests.golden import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def AG(cuda):
    from eavsr_amd import autograd as _ag, ops
    ops.lib()
    return _ag


def leaf(t, dev=None):
    t = t.clone().to(dev) if dev is not None else t.clone()
    return t.requires_grad_(True)


def grads(out, G, inputs):
    return torch.autograd.grad((out * G).sum(), inputs, allow_unused=True)


def check(gpu_grads, cpu_grads, tol, names):
    for g, c, nme in zip(gpu_grads, cpu_grads, names):
        assert (g is None) == (c is None), nme
        if c is not None:
            scale = max(1.0, c.abs().max().item())
            assert H.maxabs(g.cpu(), c) <= tol * scale, (nme, H.maxabs(g.cpu(), c), scale)


@pytest.mark.parametrize("k,chans,cout,act,res", [(3, [64], 64, "relu", False), (3, [64, 64, 64], 64, "lrelu", False),
                                                  (3, [64], 64, None, True), (1, [64, 64, 64], 64, None, False),
                                                  (5, [64], 120, None, False), (3, [18], 2, None, False),
                                                  (3, [64], 6, None, False), (3, [64], 256, "lrelu", False),
                                                  (3, [256], 64, None, False), (3, [3], 64, "relu", False)])
def test_conv2d_backward(AG, cuda, k, chans, cout, act, res):
    n, h, w = 2, 19, 37
    cin = sum(chans)
    srcs = [cases.randn(10 + i, n, c, h, w) for i, c in enumerate(chans)]
    wt = cases.randn(20, cout, cin, k, k, scale=1.0 / (cin * k * k) ** 0.5)
    b = cases.randn(21, cout, scale=0.1)
    r = cases.randn(22, n, cout, h, w) if res else None
    G = cases.randn(23, n, cout, h, w)
    cs, cw, cb = [leaf(s) for s in srcs], leaf(wt), leaf(b)
    cr = leaf(r) if res else None
    y = F.conv2d(torch.cat(cs, 1), cw, cb, 1, k // 2)
    y = F.relu(y) if act == "relu" else (F.leaky_relu(y, 0.1) if act == "lrelu" else y)
    y = y + cr if res else y
    ref = grads(y, G, cs + [cw, cb] + ([cr] if res else []))
    gs, gw, gb = [leaf(s, cuda) for s in srcs], leaf(wt, cuda), leaf(b, cuda)
    gr = leaf(r, cuda) if res else None
    out = AG.conv2d(gs, gw, gb, act=act, slope=0.1, residual=gr)
    got = grads(out, G.to(cuda), gs + [gw, gb] + ([gr] if res else []))
    check(got, ref, 3e-5, [f"src{i}" for i in range(len(chans))] + ["weight", "bias", "residual"])


def test_conv2d_multi_head_backward(AG, cuda):
    x = cases.randn(1, 1, 64, 14, 18)
    ws = [cases.randn(2 + i, co, 64, 3, 3, scale=0.05) for i, co in enumerate((4, 2))]
    bs = [cases.randn(5 + i, co, scale=0.1) for i, co in enumerate((4, 2))]
    G = cases.randn(9, 1, 6, 14, 18)
    cx, cws, cbs = leaf(x), [leaf(w_) for w_ in ws], [leaf(b_) for b_ in bs]
    ref = grads(F.conv2d(cx, torch.cat(cws), torch.cat(cbs), 1, 1), G, [cx] + cws + cbs)
    gx, gws, gbs = leaf(x, cuda), [leaf(w_, cuda) for w_ in ws], [leaf(b_, cuda) for b_ in bs]
    got = grads(AG.conv2d(gx, gws, gbs), G.to(cuda), [gx] + gws + gbs)
    check(got, ref, 3e-5, ["x", "w0", "w1", "b0", "b1"])


@pytest.mark.parametrize("shape", [(2, 64, 24, 40), (1, 2, 13, 17), (1, 5, 9, 11)])
def test_flow_warp_backward(AG, cuda, shape):
    n, c, h, w = shape
    x, f1, f2 = cases.randn(1, n, c, h, w), cases.randn(2, n, 2, h, w, scale=2.0), cases.randn(3, n, 2, h, w)
    G = cases.randn(4, n, c, h, w)
    cx, c1, c2 = leaf(x), leaf(f1), leaf(f2)
    ref = grads(O.flow_warp(cx, c1 + c2), G, [cx, c1, c2])
    gx, g1, g2 = leaf(x, cuda), leaf(f1, cuda), leaf(f2, cuda)
    got = grads(AG.flow_warp(gx, g1, flow2=g2), G.to(cuda), [gx, g1, g2])
    check(got, ref, 5e-5, ["x", "flow", "flow2"])


@pytest.mark.parametrize("sigma", [0.5, 3.0])
@pytest.mark.parametrize("shape", [(1, 64, 12, 20, 64, 8), (2, 16, 9, 12, 32, 2)])
def test_dcnv2_backward(AG, cuda, shape, sigma):
    n, c, h, w, cout, dg = shape
    x = cases.randn(1, n, c, h, w)
    off = cases.randn(2, n, dg * 18, h, w, scale=sigma)
    mask = cases.rand(3, n, dg * 9, h, w)
    wt = cases.randn(4, cout, c, 3, 3, scale=1.0 / (c * 9) ** 0.5)
    b = cases.randn(5, cout, scale=0.1)
    G = cases.randn(6, n, cout, h, w)
    cl = [leaf(t) for t in (x, off, mask, wt, b)]
    ref = grads(O.dcnv2(cl[0], cl[1], cl[2], cl[3], cl[4], 1, 1, 1, 1, dg), G, cl)
    gl = [leaf(t, cuda) for t in (x, off, mask, wt, b)]
    got = grads(AG.modulated_deform_conv2d(gl[0], gl[1], gl[2], gl[3], gl[4], 1, 1, 1, 1, dg), G.to(cuda), gl)
    check(got, ref, 1e-4, ["x", "offset", "mask", "weight", "bias"])


@pytest.mark.parametrize("shape,sigma", [((2, 96, 96), 1.0), ((2, 96, 96), 6.0), ((1, 45, 77), 2.0), ((3, 7, 5), 1.0), ((1, 4, 16), 0.3),
                                         ((2, 33, 130), 12.0)])
def test_dcnv2_backward_sampler_side_equals_the_column_path(AG, cuda, shape, sigma):
    """round 6: eavsr_dcnv2_bwd_f32 (csrc/dcn_bwd.hip: column gradient in the MFMA accumulators, in-wave reduction of d_offset / d_mask,
    dx through an LDS window, dW from re-sampled columns; no column tensor) against the rounds-1-5 path (im2col -> column tensor ->
    two GEMM launches -> col2im) on the same inputs: the training crop, ragged tiles, single tiles, offsets that leave the dx window
    (sigma = 6, 12: the direct global atomics) and the image; and the kernels that ran."""
    from eavsr_amd import ops
    n, h, w = shape
    x, off = cases.randn(501, n, 64, h, w), cases.randn(502, n, 144, h, w, scale=sigma)
    mask, wt = cases.rand(503, n, 72, h, w), cases.randn(504, 64, 64, 3, 3, scale=1.0 / 24)
    b, G = cases.randn(505, 64, scale=0.1), cases.randn(506, n, 64, h, w)
    res, names = {}, {}
    prev = ops.DCN_BWD
    try:
        for mode in ("columns", "sampler"):
            ops.DCN_BWD = mode
            gl = [leaf(t, cuda) for t in (x, off, mask, wt, b)]
            with ops.profile() as prof:
                res[mode] = grads(AG.modulated_deform_conv2d(gl[0], gl[1], gl[2], gl[3], gl[4], 1, 1, 1, 1, 8), G.to(cuda), gl)
            names[mode] = set(prof.summary())
    finally:
        ops.DCN_BWD = prev
    assert "dcnv2_bwd" in names["sampler"] and not ({"dcnv2_im2col", "dcnv2_col2im"} & names["sampler"]), names["sampler"]
    assert {"dcnv2_im2col", "dcnv2_col2im"} <= names["columns"]
    for key, a, c in zip(("dx", "doffset", "dmask", "dweight", "dbias"), res["sampler"], res["columns"]):
        sc = max(1e-6, c.abs().max().item())
        assert H.maxabs(a.cpu(), c.cpu()) <= 2e-5 * sc, (key, H.maxabs(a.cpu(), c.cpu()), sc)


@pytest.mark.parametrize("name", list(cases.G4_CASES))
def test_dcnv2_backward_committed_gradients(AG, cuda, name):
    """fp64-autograd gradients of the DCNv2 known-answer cases, stored in tests/golden/g9_gradients.npz."""
    gold = H.golden("g9_gradients")
    x, off, mask, wt, b, dg = cases.g4_inputs(name)
    gl = [leaf(t, cuda) for t in (x, off, mask, wt, b)]
    out = AG.modulated_deform_conv2d(gl[0], gl[1], gl[2], gl[3], gl[4], 1, 1, 1, 1, dg)
    got = grads(out, cases.g9_cotangent(name, out.shape).to(cuda), gl)
    for key, gr in zip(("dx", "doffset", "dmask", "dweight", "dbias"), got):
        want = gold[f"dcn_{name}__{key}"]
        diff = (gr.cpu() - want).abs()
        tol = 1e-4 * max(1.0, want.abs().max().item())
        if name == "dg8_edge" and key in ("doffset", "dx", "dweight", "dmask"):
            # positions sit exactly on the validity boundary / on integers, where d/d offset is discontinuous: a sample
            # whose position rounds to the other side in fp32 legitimately takes the other one-sided derivative
            assert (diff > tol).float().mean().item() <= 0.02, (key, (diff > tol).float().mean().item())
        else:
            assert diff.max().item() <= tol, (key, diff.max().item())


def test_flow_warp_backward_committed_gradients(AG, cuda):
    gold = H.golden("g9_gradients")
    for name, (x, flow, pad) in cases.g1_flow_warp_cases().items():
        if pad != "zeros":
            continue       # 'border' is only used inside the frozen SPyNet: it has no backward kernel (AG.flow_warp raises)
        gx, gf = leaf(x, cuda), leaf(flow, cuda)
        out = AG.flow_warp(gx, gf, padding_mode=pad)
        dx, dflow = grads(out, cases.g9_cotangent(name, out.shape).to(cuda), [gx, gf])
        for key, gr in (("dx", dx), ("dflow", dflow)):
            if (name, key) == ("c4_int_zeros", "dflow"):
                continue   # exact-integer positions: one-sided derivatives, see tests/test_oracle_dcn.py
            want = gold[f"warp_{name}__{key}"]
            bad = ((gr.cpu() - want).abs() > 2e-4 * max(1.0, want.abs().max().item())).float().mean().item()
            assert bad <= 0.002, (name, key, bad)


def test_adapt_frontend_and_affine_backward(AG, cuda):
    sd = H.filled(H.adaptoffset_shapes("f."), "trained_like")
    x, hh = cases.randn(1, 2, 64, 11, 14), cases.randn(2, 2, 64, 11, 14)
    keys = ["f.concat.0.weight", "f.concat.0.bias", "f.concat2.0.weight", "f.concat2.0.bias"]
    G = cases.randn(3, 2, 64, 11, 14)
    cp = {k: leaf(sd[k]) for k in keys}
    cx, ch = leaf(x), leaf(hh)
    ref = grads(O.adapt_frontend({**sd, **cp}, "f.", cx, ch), G, [cx, ch] + [cp[k] for k in keys])
    gp = {k: leaf(sd[k], cuda) for k in keys}
    gx, gh = leaf(x, cuda), leaf(hh, cuda)
    got = grads(AG.adapt_frontend(gx, gh, *[gp[k] for k in keys]), G.to(cuda), [gx, gh] + [gp[k] for k in keys])
    check(got, ref, 3e-5, ["x", "h"] + keys)
    # affine -> offsets (+ mask)
    heads = cases.randn(4, 2, 120, 9, 13)
    Go, Gm = cases.randn(5, 2, 144, 9, 13), cases.randn(6, 2, 72, 9, 13)
    chd = leaf(heads)
    ref = torch.autograd.grad((O.affine_offsets(chd[:, :32], chd[:, 32:48], 8) * Go).sum() +
                              (torch.sigmoid(chd[:, 48:]) * Gm).sum(), [chd])
    ghd = leaf(heads, cuda)
    off, mask = AG.affine_offsets(ghd, 8, True)
    got = torch.autograd.grad((off * Go.to(cuda)).sum() + (mask * Gm.to(cuda)).sum(), [ghd])
    check(got, ref, 2e-5, ["heads"])


def test_resampling_backward(AG, cuda):
    x, pre = cases.randn(1, 2, 2, 8, 12), cases.randn(2, 2, 2, 8, 12)
    post, G = cases.randn(3, 2, 2, 16, 24), cases.randn(4, 2, 2, 16, 24)
    cl = [leaf(t) for t in (x, pre, post)]
    ref = grads(F.interpolate(cl[0] + cl[1], size=(16, 24), mode="bilinear", align_corners=True) * 2.0 + cl[2], G, cl)
    gl = [leaf(t, cuda) for t in (x, pre, post)]
    got = grads(AG.resize_bilinear_ac(gl[0], (16, 24), 2.0, pre_add=gl[1], post_add=gl[2]), G.to(cuda), gl)
    check(got, ref, 2e-5, ["x", "pre", "post"])
    f = cases.randn(5, 3, 64, 12, 20)
    G2, G4 = cases.randn(6, 3, 64, 6, 10), cases.randn(7, 3, 64, 3, 5)
    cf = leaf(f)
    d2, d4 = O.feature_pyramid(cf)
    ref = torch.autograd.grad((d2 * G2).sum() + (d4 * G4).sum(), [cf])
    gf = leaf(f, cuda)
    e2, e4 = AG.pyramid(gf)
    got = torch.autograd.grad((e2 * G2.to(cuda)).sum() + (e4 * G4.to(cuda)).sum(), [gf])
    check(got, ref, 1e-6, ["x"])


@pytest.mark.parametrize("hw", [(10, 14), (9, 7), (96, 96)])
def test_rcab_tail_backward(AG, cuda, hw):
    """the RCAB tail's backward (plane sums + eavsr_rcab_tail_bwd_f32: MLP backward, mean broadcast and dr in one launch) against
    CPU autograd of the oracle's CALayer; float4 and ragged planes; then two uses inside grad_sink (the launch ADDS the
    parameter gradients of the second use in place) against the sum of two autograd runs"""
    sd = H.filled(H.rcab_shapes("b."), "trained_like")
    keys = ["b.ca.conv_du.0.weight", "b.ca.conv_du.0.bias", "b.ca.conv_du.2.weight", "b.ca.conv_du.2.bias"]
    r, x, G = cases.randn(1, 2, 64, *hw), cases.randn(2, 2, 64, *hw), cases.randn(3, 2, 64, *hw)
    cp = {k: leaf(sd[k]) for k in keys}
    cr, cx = leaf(r), leaf(x)
    ref = grads(O.ca_layer({**sd, **cp}, "b.ca.", cr) + cx, G, [cr, cx] + [cp[k] for k in keys])
    gp = {k: leaf(sd[k], cuda) for k in keys}
    gr, gx = leaf(r, cuda), leaf(x, cuda)
    got = grads(AG.rcab_tail(gr, gx, *[gp[k] for k in keys]), G.to(cuda), [gr, gx] + [gp[k] for k in keys])
    check(got, ref, 2e-5, ["r", "x"] + keys)
    # two uses of the same parameters under grad_sink: out = tail(tail(r, x), x)
    cp2 = {k: leaf(sd[k]) for k in keys}
    cr2, cx2 = leaf(r), leaf(x)
    y1 = O.ca_layer({**sd, **cp2}, "b.ca.", cr2) + cx2
    ref2 = grads(O.ca_layer({**sd, **cp2}, "b.ca.", y1) + cx2, G, [cr2, cx2] + [cp2[k] for k in keys])
    gp2 = {k: torch.nn.Parameter(sd[k].to(cuda)) for k in keys}
    gr2, gx2 = leaf(r, cuda), leaf(x, cuda)
    with AG.grad_sink():
        y = AG.rcab_tail(AG.rcab_tail(gr2, gx2, *[gp2[k] for k in keys]), gx2, *[gp2[k] for k in keys])
        (y * G.to(cuda)).sum().backward()
    got2 = [gr2.grad, gx2.grad] + [gp2[k].grad for k in keys]
    check(got2, ref2, 5e-5, ["r", "x"] + keys)


def _grads_of_module(mod, sd, prefix, loss_fn):
    loss = loss_fn()
    names = [k for k, p in mod.named_parameters() if p.requires_grad]
    gs = torch.autograd.grad(loss, [p for _, p in mod.named_parameters() if p.requires_grad], allow_unused=True)
    return dict(zip(names, gs))


@pytest.mark.parametrize("preset", ["trained_like"])
def test_multiadstn_backward_vs_oracle(AG, cuda, preset):
    from eavsr_amd import networks as Nw
    opt = Namespace(predict=False, n_frame=7, n_flow=5, scale=4)
    sd = H.filled(H.multiadstn_shapes("a."), preset)
    nbr, ref, fp, flow = cases.g5_inputs(h=16, w=24)
    G = cases.randn(99, 1, 64, 16, 24)
    # CPU oracle autograd
    csd = {k: (leaf(v) if not k.endswith("regular_matrix") else v) for k, v in sd.items()}
    cfp = leaf(fp)
    out_c = O.multi_adstn(csd, "a.", nbr, ref, cfp, flow, 8)
    pkeys = [k for k in csd if not k.endswith("regular_matrix")]
    ref_g = torch.autograd.grad((out_c * G).sum(), [csd[k] for k in pkeys] + [cfp], allow_unused=True)
    # HIP
    m = Nw.MultiAdSTN(opt, 64, 64, deformable_groups=8)
    m.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=True)
    m = m.to(cuda).train()
    gfp = leaf(fp, cuda)
    out_g = m([t.to(cuda) for t in nbr], [t.to(cuda) for t in ref], gfp, flow.to(cuda))
    assert H.maxabs(out_g.detach().cpu(), out_c.detach()) <= 1e-4
    params = dict(m.named_parameters())
    got = torch.autograd.grad((out_g * G.to(cuda)).sum(), [params[k[2:]] for k in pkeys] + [gfp], allow_unused=True)
    bad = []
    for k, g, c in zip(pkeys + ["feat_prop"], got, ref_g):
        if c is None:
            assert g is None or g.abs().max().item() == 0, k
            continue
        scale = max(1e-3, c.abs().max().item())
        if H.maxabs(g.cpu(), c) > 2e-3 * scale:
            bad.append((k, H.maxabs(g.cpu(), c), scale))
    assert not bad, bad


def test_training_step_decreases_loss_and_matches_oracle_grads(AG, cuda):
    """One EAVSRP x4 training step at 1 x 3 x 3 x 64 x 64: loss / a sample of parameter gradients against CPU
    autograd through the oracle, then an Adam step lowers the L1 loss."""
    from eavsr_amd.eavsrp_model import EAVSRP
    from eavsr_amd.utils.synthetic import synthetic_clip
    opt = Namespace(predict=False, n_frame=7, n_flow=5, scale=4)
    sd = H.filled(H.model_shapes("x4"), "trained_like")
    net = EAVSRP(opt, None)
    net.load_state_dict(sd, strict=True)
    net = net.to(cuda).train()
    clip = synthetic_clip(1, 3, 64, 64, seed=11)
    hr = synthetic_clip(1, 3, 256, 256, seed=12)
    out = net(clip.to(cuda))
    loss = (out - hr.to(cuda)).abs().mean()
    loss.backward()
    # oracle gradients for a few parameters spread over the path
    watch = ["conv_last.weight", "backbone.forward_2.main.2.rg.3.res.0.weight", "fusion.backward_1.weight",
             "deform_align.forward_1.weight", "deform_align.backward_2.adastn.mask_conv.bias",
             "deform_align.forward_1.flow_l2.concat.0.weight", "encoder.tail.bias",
             "backbone.backward_1.main.2.rg.7.ca.conv_du.0.weight"]
    csd = {k: (v.clone().requires_grad_(True) if k in watch else v) for k, v in sd.items()}
    with torch.enable_grad():
        with torch.no_grad():
            flows = O.compute_flow(sd, clip)
        out_c = _oracle_forward_with_grad(csd, clip, flows)
        loss_c = (out_c - hr).abs().mean()
    gs = torch.autograd.grad(loss_c, [csd[k] for k in watch])
    assert abs(loss.item() - loss_c.item()) <= 1e-4
    params = dict(net.named_parameters())
    for k, gc in zip(watch, gs):
        gg = params[k].grad
        assert gg is not None, k
        scale = max(1e-7, gc.abs().max().item())
        assert H.maxabs(gg.cpu(), gc) <= 5e-3 * scale, (k, H.maxabs(gg.cpu(), gc), scale)
    assert all(p.grad is None for p in net.spynet.parameters())
    optim = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=1e-4)
    optim.step()
    with torch.no_grad():
        loss2 = (net(clip.to(cuda)) - hr.to(cuda)).abs().mean()
    assert loss2.item() < loss.item()


def _oracle_forward_with_grad(sd, lrs, flows):
    """eavsrp_forward without its no_grad wrapper"""
    n, t, c, h, w = lrs.shape
    ff, fb = flows
    f1 = O.encoder(sd, "encoder.", lrs.reshape(-1, c, h, w))
    d2, d4 = O.feature_pyramid(f1)
    f1, d2, d4 = f1.view(n, t, -1, h, w), d2.view(n, t, -1, h // 2, w // 2), d4.view(n, t, -1, h // 4, w // 4)
    feats = {"spatial": [f1[:, i] for i in range(t)], "spatial_d2": [d2[:, i] for i in range(t)],
             "spatial_d4": [d4[:, i] for i in range(t)]}
    for it in (1, 2):
        for direction in ("backward", "forward"):
            module = f"{direction}_{it}"
            feats[module] = []
            feats = O.propagate(sd, feats, fb if direction == "backward" else ff, module, 8)
    return O.upsample(sd, lrs, feats, 4)


def test_model_wrapper_training_step(AG, cuda):
    """EAVSRPModel.optimize_parameters: two Adam groups (alignment modules at lr 1e-5), L1 loss, parameters move."""
    from eavsr_amd.eavsrp_model import EAVSRPModel
    from eavsr_amd.utils.synthetic import synthetic_clip
    opt = Namespace(predict=False, n_frame=3, n_flow=5, scale=4, isTrain=True, gpu_ids=[0], lr=1e-4, beta1=0.9,
                    beta2=0.999, weight_decay=0.0, npost=350, lr_policy="step", lr_decay_iters=1, load_iter=0, load_path="",
                    load_optimizers=False, verbose=False)
    model = EAVSRPModel(opt)
    model.setup(opt)                       # what train_basic.py:42 calls: schedulers (no checkpoint named: nothing to load)
    sd = H.filled(H.model_shapes("x4"), "trained_like")
    model.netEAVSRP.load_state_dict(sd, strict=True)
    groups = model.optimizer_EAVSRP.param_groups
    assert groups[0]["lr"] == 1e-4 and groups[1]["lr"] == 1e-5
    n_align = sum(p.numel() for p in model.netEAVSRP.deform_align.parameters())
    assert sum(p.numel() for p in groups[1]["params"]) == n_align
    # the groups hold ALL 13,718,099 parameters, as the reference builds them (eavsrp_model.py:45-59); 12,277,799 of them are
    # trainable (1,440,300 frozen SPyNet parameters never get a gradient, so Adam never touches them)
    assert sum(p.numel() for g_ in groups for p in g_["params"]) == 13718099
    assert sum(p.numel() for g_ in groups for p in g_["params"] if p.requires_grad) == 12277799
    data = {"lr_seq": synthetic_clip(1, 3, 64, 64, seed=1), "hr_seq": synthetic_clip(1, 3, 256, 256, seed=2), "fname": "x"}
    before = model.netEAVSRP.conv_last.weight.detach().clone()
    losses = []
    for it in range(3):
        model.set_input(data, epoch=0)
        model.optimize_parameters()
        losses.append(model.get_current_losses()["EAVSRP_L1"])
    assert losses[-1] < losses[0]
    assert not torch.equal(before, model.netEAVSRP.conv_last.weight.detach())
    model.update_learning_rate()           # train_basic.py:80: StepLR(step_size=1, gamma=0.5) halves both groups
    assert [g_["lr"] for g_ in groups] == pytest.approx([5e-5, 5e-6])
    model.eval()
    model.set_input(data)
    model.test()
    assert tuple(model.data_sr_seq.shape) == (1, 3, 3, 256, 256)


def test_checkpoint_optimizer_files_and_eval_harness(AG, cuda, tmp_path):
    """`<name>_model_<epoch>.pth` = {'state_dict'}, `<optimizer name>.pth` = {'name','epoch',
    'state_dict'} (base_model.py:159-270), strict loading, resume reproduces the next step; the test
    loop of test_basic.py on synthetic items."""
    from eavsr_amd.eavsrp_model import EAVSRPModel
    from eavsr_amd import harness
    from eavsr_amd.utils.synthetic import synthetic_clip
    mk = lambda: Namespace(predict=False, n_frame=3, n_flow=5, scale=4, isTrain=True, gpu_ids=[0], lr=1e-4, beta1=0.9,
                           beta2=0.999, weight_decay=0.0, npost=350, checkpoints_dir=str(tmp_path), name="run",
                           optimizer="Adam", load_path="")
    data = {"lr_seq": synthetic_clip(1, 3, 64, 64, seed=1), "hr_seq": synthetic_clip(1, 3, 256, 256, seed=2), "fname": "x"}
    model = EAVSRPModel(mk())
    model.netEAVSRP.load_state_dict(H.filled(H.model_shapes("x4"), "trained_like"), strict=True)
    for _ in range(2):
        model.set_input(data, epoch=0)
        model.optimize_parameters()
    path = model.save_networks(7)
    assert path.endswith("run/EAVSRP_model_7.pth") and H.os.path.exists(H.os.path.join(str(tmp_path), "run", "EAVSRP_optimizer_Adam.pth"))
    blob = torch.load(path, map_location="cpu")
    assert set(blob) == {"state_dict"} and set(blob["state_dict"]) == set(model.netEAVSRP.state_dict())
    ob = torch.load(H.os.path.join(str(tmp_path), "run", "EAVSRP_optimizer_Adam.pth"), map_location="cpu")
    assert ob["name"] == "EAVSRP_optimizer_Adam" and ob["epoch"] == 7 and len(ob["state_dict"]["param_groups"]) == 2
    model.set_input(data, epoch=0)
    model.optimize_parameters()          # step 3 of the original run
    want = {k: v.detach().clone() for k, v in model.netEAVSRP.state_dict().items()}

    resumed = EAVSRPModel(mk())
    resumed.load_networks(7)
    resumed.load_optimizers(7)
    assert resumed.start_epoch == 7
    resumed.set_input(data, epoch=0)
    resumed.optimize_parameters()
    got = resumed.netEAVSRP.state_dict()
    worst = max((got[k].float() - want[k].float()).abs().max().item() for k in want)
    assert worst < 1e-6, worst           # restored Adam moments: the same step (float atomics in the backward scatter
                                         # kernels make it reproducible to rounding, not bit for bit)
    with pytest.raises(RuntimeError):
        resumed.load_optimizers(8)       # epoch recorded in the file must match
    bad = dict(blob["state_dict"])
    bad.pop("conv_last.bias")
    torch.save({"state_dict": bad}, H.os.path.join(str(tmp_path), "bad.pth"))
    with pytest.raises(RuntimeError):
        resumed.load_networks(H.os.path.join(str(tmp_path), "bad.pth"))
    bad["conv_last.bias"] = blob["state_dict"]["conv_last.bias"]
    bad["stray.weight"] = torch.zeros(1)
    torch.save({"state_dict": bad}, H.os.path.join(str(tmp_path), "bad2.pth"))
    with pytest.raises(RuntimeError):
        resumed.load_networks(H.os.path.join(str(tmp_path), "bad2.pth"))

    items = [dict(data), dict(data)]
    rep = harness.evaluate(resumed, items)
    assert rep["frames"] == 6 and len(rep["psnr"]) == 2 and rep["psnr"][0] == rep["psnr"][1]
    vis = resumed.get_current_visuals()
    assert abs(rep["psnr"][0] - harness.calc_psnr(vis["data_sr_seq"], vis["data_hr_seq"])) < 1e-6
    assert rep["frames_per_s"] > 0 and resumed.num == 1    # the wrapper skips its first timed call (eavsrp_model.py:104-107)


def _dp_worker(rank, world, port, q):
    import os
    import sys
    sys.path.insert(0, H.os.path.dirname(H.os.path.dirname(H.os.path.abspath(H.__file__))))
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), EAVSR_DIST_BACKEND="gloo")
    from eavsr_amd import shard
    from eavsr_amd.eavsrp_model import EAVSRP
    from eavsr_amd.utils.synthetic import synthetic_clip
    shard.init_process_group()
    dev = torch.device("cuda:0")   # both ranks share the one GPU of the test box (gloo moves the buckets)
    net = EAVSRP(Namespace(predict=False, n_frame=3, n_flow=5, scale=4), None)
    net.load_state_dict(H.filled(H.model_shapes("x4"), "trained_like"), strict=True)
    net = net.to(dev).train()
    sync = shard.GradientAllReducer(net.parameters())
    clips, hrs = synthetic_clip(world, 3, 64, 64, seed=21), synthetic_clip(world, 3, 256, 256, seed=22)
    mine = shard.clip_indices(world, rank, world)
    out = net(clips[mine].to(dev))
    (out - hrs[mine].to(dev)).abs().mean().backward()
    sync.finish()
    names = ["conv_last.weight", "deform_align.backward_1.weight", "backbone.forward_1.main.0.bias"]
    params = dict(net.named_parameters())
    q.put((rank, {k: params[k].grad.cpu().numpy() for k in names}))   # arrays travel by value (no shared-memory handle)
    shard.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_data_parallel_gradients_equal_the_mean_of_per_clip_gradients(AG, cuda):
    """SURVEY 8e, training: clips sharded over ranks, ONE all-reduce on the loss gradients; the synchronised
    gradient equals the mean of the single-clip gradients computed in this process."""
    import socket
    import torch.multiprocessing as mp
    from eavsr_amd.eavsrp_model import EAVSRP
    from eavsr_amd.utils.synthetic import synthetic_clip
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r_: {k: torch.from_numpy(v) for k, v in d_.items()} for r_, d_ in (q.get(timeout=600) for _ in range(2))}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    net = EAVSRP(Namespace(predict=False, n_frame=3, n_flow=5, scale=4), None)
    net.load_state_dict(H.filled(H.model_shapes("x4"), "trained_like"), strict=True)
    net = net.to(cuda).train()
    clips, hrs = synthetic_clip(2, 3, 64, 64, seed=21), synthetic_clip(2, 3, 256, 256, seed=22)
    acc = {}
    params = dict(net.named_parameters())
    for i in range(2):
        net.zero_grad(set_to_none=True)
        (net(clips[i:i + 1].to(cuda)) - hrs[i:i + 1].to(cuda)).abs().mean().backward()
        for k in res[0]:
            acc[k] = acc.get(k, 0) + params[k].grad.cpu() / 2
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k
        scale = max(1e-8, acc[k].abs().max().item())
        assert H.maxabs(res[0][k], acc[k]) <= 2e-3 * scale, (k, H.maxabs(res[0][k], acc[k]), scale)


@pytest.mark.parametrize("n,c,h,w", [(2, 64, 96, 96), (3, 5, 19, 37), (1, 120, 8, 4), (0, 7, 4, 4)])
def test_channel_and_plane_sums(cuda, n, c, h, w):
    """bias-gradient reduction (sum over n, h, w) and the per-plane sums of the RCAB tail, both the float4 and the
    ragged path, against float64 sums"""
    from eavsr_amd import ops
    a, b = cases.randn(70, n, c, h, w), cases.randn(71, n, c, h, w)
    got = ops.channel_sum(a.to(cuda)).cpu()
    want = a.double().sum(dim=(0, 2, 3))
    assert got.shape == (c,)
    assert H.maxabs(got.double(), want) <= 1e-5 * max(1.0, want.abs().max().item() if n else 1.0) + 1e-4
    if n:
        for bb in (None, b):
            got = ops.plane_sum(a.to(cuda), None if bb is None else bb.to(cuda), 0.5).cpu()
            want = 0.5 * (a.double() * (1 if bb is None else bb.double())).sum(dim=(2, 3))
            assert H.maxabs(got.double(), want) <= 1e-4


@pytest.mark.parametrize("n,h,w", [(4, 64, 64), (2, 19, 37), (1, 180, 320)])
def test_conv2d_relu_mask_epilogue_equals_conv_then_act_bwd(cuda, n, h, w):
    """act="relu_mask" (EAVSR_ACT_RELU_MASK): the ReLU's backward mask inside the input-gradient convolution of RCABlock's
    second conv (networks.py:461-462) -- bit-identical to the convolution followed by act_bwd, on both direct kernels"""
    from eavsr_amd import ops
    g = cases.randn(70, n, 64, h, w).to(cuda)
    t = torch.relu(cases.randn(71, n, 64, h, w)).to(cuda)       # a ReLU's forward output: exact zeros where it clipped
    wt = cases.randn(72, 64, 64, 3, 3, scale=1.0 / 24).to(cuda)
    b = cases.randn(73, 64, scale=0.1).to(cuda)
    for bias in (None, b):
        with ops.modes(conv="direct"):       # the fp32-MFMA direct kernels: the same sum with and without the mask, bit for bit
            want = ops.act_bwd(ops.conv2d(g, wt, bias), t, "relu", 0.0)
            got = ops.conv2d(g, wt, bias, act="relu_mask", residual=t)
        assert torch.equal(got, want)
        assert (got[t == 0] == 0).all()
        with ops.profile() as prof:          # the default mode: the small-launch bf16x6 kernel has the same epilogue
            got6 = ops.conv2d(g, wt, bias, act="relu_mask", residual=t)
        assert list(prof.summary()) == ["conv3x3_64to64_x6s"]
        assert (got6[t == 0] == 0).all()
        assert H.maxabs(got6.cpu(), want.cpu()) <= 2e-5 * max(1.0, want.abs().max().item())
    with pytest.raises(ValueError):
        ops.conv2d(g, wt, None, act="relu_mask")
    with pytest.raises(ValueError):
        ops.conv2d(g, wt, None, act="relu_mask", residual=t, chan_partial=True)


@pytest.mark.parametrize("k,nseg,shape", [(3, 7, (2, 64, 96, 96)), (3, 3, (1, 64, 19, 37)), (1, 8, (2, 64, 24, 40)), (5, 2, (1, 64, 16, 32)),
                                          (3, 2, (1, 64, 18, 40)), (3, 2, (3, 40, 7, 4)), (3, 8, (1, 8, 45, 80))])
def test_conv_wgrad_over_several_uses_in_one_launch(cuda, k, nseg, shape):
    """eavsr_conv_wgrad_multi_f32 / eavsr_channel_sum_multi_f32 (ABI 26): the uses of one weight across the frames of the
    recurrence as segments of ONE launch, against the sum of the per-use gradients in float64 (torch CPU) and against the
    per-use launches accumulated one after the other; two sources (a virtual concatenation), `accumulate` on top.
    3x3 with w % 4 == 0 runs the bf16x6 kernel (ABI 27: 4 x 32-pixel tiles cut by the image in both directions, fewer than 32
    valid input channels in a quadrant), 19 x 37 the fp32-MFMA kernel."""
    from eavsr_amd import ops
    n, c, h, w = shape
    cout = 64 if k != 5 else 120
    dys = [cases.randn(300 + i, n, cout, h, w) for i in range(nseg)]
    xa = [cases.randn(400 + i, n, c, h, w) for i in range(nseg)]
    xb = [cases.randn(500 + i, n, 24, h, w) for i in range(nseg)]
    want = torch.zeros(cout, c + 24, k, k, dtype=torch.float64)
    for d, a, b in zip(dys, xa, xb):
        want += torch.nn.grad.conv2d_weight(torch.cat([a, b], 1).double(), (cout, c + 24, k, k), d.double(), padding=k // 2)
    want_b = sum(d.double().sum(dim=(0, 2, 3)) for d in dys)
    gd, ga, gb = [d.to(cuda) for d in dys], [a.to(cuda) for a in xa], [b.to(cuda) for b in xb]
    out = torch.empty(cout, c + 24, k, k, device=cuda)
    ops.conv_wgrad_multi(gd, [[a, b] for a, b in zip(ga, gb)], k, out=out)
    one = torch.empty_like(out)
    for i in range(nseg):
        ops.conv_wgrad(gd[i], [ga[i], gb[i]], k, out=one, accumulate=i > 0)
    scale = max(1.0, want.abs().max().item())
    assert H.maxabs(out.cpu().double(), want) <= 2e-5 * scale
    assert H.maxabs(out.cpu(), one.cpu()) <= 2e-5 * scale
    ops.conv_wgrad_multi(gd[:1], [[ga[0], gb[0]]], k, out=out, accumulate=True)       # (+)= one more use
    want1 = want + torch.nn.grad.conv2d_weight(torch.cat([xa[0], xb[0]], 1).double(), (cout, c + 24, k, k), dys[0].double(), padding=k // 2)
    assert H.maxabs(out.cpu().double(), want1) <= 2e-5 * scale
    db2 = torch.full((cout,), 7.0, device=cuda)      # the bias gradient riding in the weight-gradient launch (ABI 27)
    out2 = torch.empty_like(out)
    ops.conv_wgrad_multi(gd, [[a, b] for a, b in zip(ga, gb)], k, out=out2, bias_out=db2)
    assert H.maxabs(out2.cpu().double(), want) <= 2e-5 * scale
    assert H.maxabs(db2.cpu().double(), want_b) <= 1e-5 * max(1.0, want_b.abs().max().item()) + 1e-4
    ops.conv_wgrad_multi(gd[:2], [[ga[0], gb[0]], [ga[1], gb[1]]], k, out=out2, bias_out=db2, accumulate=True)
    assert H.maxabs(db2.cpu().double(), want_b + dys[0].double().sum(dim=(0, 2, 3)) + dys[1].double().sum(dim=(0, 2, 3))) <= 1e-3
    db = torch.empty(cout, device=cuda)
    ops.channel_sum_multi(gd, out=db)
    assert H.maxabs(db.cpu().double(), want_b) <= 1e-5 * max(1.0, want_b.abs().max().item()) + 1e-4
    ops.channel_sum_multi(gd[:2], out=db, accumulate=True)
    assert H.maxabs(db.cpu().double(), want_b + dys[0].double().sum(dim=(0, 2, 3)) + dys[1].double().sum(dim=(0, 2, 3))) <= 1e-3
    with pytest.raises(ValueError):
        ops.conv_wgrad_multi([gd[0]] * 9, [[ga[0], gb[0]]] * 9, k, out=out)      # > 8 segments
    with pytest.raises(ValueError):
        ops.conv_wgrad_multi(gd[:2], [[ga[0], gb[0]], [ga[1][:, :c // 2], gb[1]]], k, out=out)      # shapes differ


def test_grad_sink_matches_autograd_accumulation(AG, cuda):
    """In-place accumulation of the per-use parameter gradients (autograd.grad_sink, what optimize_parameters runs)
    against autograd's own sum of the same per-use gradients, for every trainable parameter."""
    from eavsr_amd.eavsrp_model import EAVSRP
    from eavsr_amd.utils.synthetic import synthetic_clip
    opt = Namespace(predict=False, n_frame=7, n_flow=5, scale=4)
    net = EAVSRP(opt, None)
    net.load_state_dict(H.filled(H.model_shapes("x4"), "trained_like"), strict=True)
    net = net.to(cuda).train()
    clip, hr = synthetic_clip(1, 4, 64, 64, seed=21).to(cuda), synthetic_clip(1, 4, 256, 256, seed=22).to(cuda)
    (net(clip) - hr).abs().mean().backward()
    plain = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    net.zero_grad(set_to_none=True)
    with AG.grad_sink():
        (net(clip) - hr).abs().mean().backward()
    sunk = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    assert set(plain) == set(sunk) and len(plain) > 300
    for k in plain:
        scale = max(1e-8, plain[k].abs().max().item())
        assert H.maxabs(sunk[k], plain[k]) <= 2e-4 * scale, (k, H.maxabs(sunk[k], plain[k]), scale)
    # second backward: the sink now knows every call site's use count and launches at the LAST expected use instead of at flush()
    # (single-use convolutions at once: ADVICE r5) -- same sums; then a clip of another length, where the expectation is wrong
    # in both directions (a launch that comes early is followed by accumulating ones)
    assert AG.grad_sink._last_uses and 1 in set(AG.grad_sink._last_uses.values())
    net.zero_grad(set_to_none=True)
    with AG.grad_sink() as sink:
        loss = (net(clip) - hr).abs().mean()
        loss.backward()
        assert all(not e.get("pending") for e in sink.entries.values())      # nothing waits for flush(): every expected use came
    for k, p in net.named_parameters():
        if p.grad is not None:
            scale = max(1e-8, plain[k].abs().max().item())
            assert H.maxabs(p.grad, plain[k]) <= 2e-4 * scale, (k, H.maxabs(p.grad, plain[k]), scale)
    for t2 in (6, 3):
        clip2, hr2 = synthetic_clip(1, t2, 64, 64, seed=23).to(cuda), synthetic_clip(1, t2, 256, 256, seed=24).to(cuda)
        net.zero_grad(set_to_none=True)
        (net(clip2) - hr2).abs().mean().backward()
        plain2 = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
        net.zero_grad(set_to_none=True)
        with AG.grad_sink():
            (net(clip2) - hr2).abs().mean().backward()
        for k, p in net.named_parameters():
            if p.grad is not None:
                scale = max(1e-8, plain2[k].abs().max().item())
                assert H.maxabs(p.grad, plain2[k]) <= 2e-4 * scale, (t2, k, H.maxabs(p.grad, plain2[k]), scale)
    # outside the context nothing is redirected, and a second context cannot be nested
    with AG.grad_sink():
        with pytest.raises(RuntimeError):
            AG.grad_sink().__enter__()


def test_graphed_training_step_matches_eager(AG, cuda):
    """graph.GraphedTrainStep: one captured HIP graph of forward + L1 + backward + Adam replays to the same losses
    and parameters as eager optimize_parameters (float atomics in two scatter kernels: equal to rounding)."""
    from eavsr_amd.eavsrp_model import EAVSRPModel
    from eavsr_amd.graph import GraphedTrainStep
    from eavsr_amd.utils.synthetic import synthetic_clip
    mk = lambda: Namespace(predict=False, n_frame=3, n_flow=5, scale=4, isTrain=True, gpu_ids=[0], lr=1e-4, beta1=0.9,
                           beta2=0.999, weight_decay=0.0, npost=350)
    sd = H.filled(H.model_shapes("x4"), "trained_like")
    data = {"lr_seq": synthetic_clip(1, 3, 64, 64, seed=1), "hr_seq": synthetic_clip(1, 3, 256, 256, seed=2), "fname": "x"}
    data2 = {"lr_seq": synthetic_clip(1, 3, 64, 64, seed=3), "hr_seq": synthetic_clip(1, 3, 256, 256, seed=4), "fname": "y"}
    eager = EAVSRPModel(mk())
    eager.netEAVSRP.load_state_dict(sd, strict=True)
    want = []
    for d in (data, data, data2, data):
        eager.set_input(d, epoch=0)
        eager.optimize_parameters()
        want.append(eager.get_current_losses()["EAVSRP_L1"])
    graphed = EAVSRPModel(mk())
    graphed.netEAVSRP.load_state_dict(sd, strict=True)
    graphed.set_input(data, epoch=0)
    g = GraphedTrainStep(graphed, warmup=1)           # one eager step on `data`, then the capture (not a step)
    got = []
    for d in (data, data2, data):
        g.step({k: v.to(cuda) for k, v in d.items() if k != "fname"})
        got.append(graphed.get_current_losses()["EAVSRP_L1"])
    assert all(abs(a - b) <= 2e-5 * max(1.0, abs(b)) for a, b in zip(got, want[1:])), (got, want)
    pe, pg = dict(eager.netEAVSRP.named_parameters()), dict(graphed.netEAVSRP.named_parameters())
    worst = max((pe[k].detach() - pg[k].detach()).abs().max().item() for k in pe)
    assert worst <= 2.5e-4, worst      # Adam moves a parameter by ~lr = 1e-4 per step whatever the gradient's size
    with pytest.raises(ValueError):
        g.step({"lr_seq": torch.zeros(1, 3, 3, 32, 32, device=cuda), "hr_seq": torch.zeros(1, 3, 3, 128, 128, device=cuda)})
    # eager evaluation after replays sees the replayed parameters (no stale packed weights): the same output as a
    # fresh model loaded from the replayed parameters
    fresh = EAVSRPModel(mk())
    fresh.netEAVSRP.load_state_dict({k: v.detach().clone() for k, v in graphed.netEAVSRP.state_dict().items()}, strict=True)
    graphed.eval(); fresh.eval()
    graphed.set_input(data); fresh.set_input(data)
    graphed.test(); fresh.test()
    assert H.maxabs(graphed.data_sr_seq.cpu(), fresh.data_sr_seq.cpu()) <= 1e-6


def test_eager_calls_between_replays_never_see_stale_derived_weights(AG, cuda):
    """a replayed graph updates the parameters without bumping `_version`, so every cache of a derived weight form
    (packed, Winograd-transformed, bf16-split, transposed for the input gradient, flipped for DCNv2's backward) must be dropped
    around a replay -- ALL of them register in ops.WEIGHT_CACHES.  Eager forward + backward (no optimizer step), replays, then
    the same eager forward + backward at the unchanged `_version`: outputs and gradients equal those of a fresh model loaded
    from the replayed parameters."""
    from eavsr_amd import ops
    from eavsr_amd.eavsrp_model import EAVSRPModel
    from eavsr_amd.graph import GraphedTrainStep
    from eavsr_amd.utils.synthetic import synthetic_clip
    mk = lambda: Namespace(predict=False, n_frame=3, n_flow=5, scale=4, isTrain=True, gpu_ids=[0], lr=1e-3, beta1=0.9,
                           beta2=0.999, weight_decay=0.0, npost=350)
    sd = H.filled(H.model_shapes("x4"), "trained_like")
    data = {"lr_seq": synthetic_clip(1, 3, 64, 64, seed=1), "hr_seq": synthetic_clip(1, 3, 256, 256, seed=2), "fname": "x"}
    lr, hr = data["lr_seq"].to(cuda), data["hr_seq"].to(cuda)
    m = EAVSRPModel(mk())
    m.netEAVSRP.load_state_dict(sd, strict=True)
    m.set_input(data, epoch=0)
    g = GraphedTrainStep(m, warmup=1)
    watch = ["backbone.forward_1.main.2.rg.3.res.0.weight", "deform_align.backward_1.weight", "fusion.forward_2.weight",
             "deform_align.forward_1.adastn.mask_conv.weight", "conv_last.weight"]

    def eager_fwd_bwd(net):
        for p in net.parameters():
            p.grad = None
        with AG.grad_sink():
            out = net(lr)
            (out - hr).abs().mean().backward()
        prm = dict(net.named_parameters())
        return out.detach().clone(), {k: prm[k].grad.detach().clone() for k in watch}

    g.step()
    versions = {k: p._version for k, p in m.netEAVSRP.named_parameters()}
    eager_fwd_bwd(m.netEAVSRP)                               # fills every derived-weight cache at this version
    assert any(len(getattr(d, "_d", d)) for d in ops.WEIGHT_CACHES)
    for _ in range(3):
        g.step()                                             # parameters move on the device; versions do not
    assert {k: p._version for k, p in m.netEAVSRP.named_parameters()} == versions
    out, grads = eager_fwd_bwd(m.netEAVSRP)
    fresh = EAVSRPModel(mk())
    fresh.netEAVSRP.load_state_dict({k: v.detach().clone() for k, v in m.netEAVSRP.state_dict().items()}, strict=True)
    fresh.netEAVSRP.train()
    out_f, grads_f = eager_fwd_bwd(fresh.netEAVSRP)
    assert H.maxabs(out.cpu(), out_f.cpu()) <= 1e-6
    for k in watch:
        sc = max(1e-7, grads_f[k].abs().max().item())
        assert H.maxabs(grads[k].cpu(), grads_f[k].cpu()) <= 1e-4 * sc, (k, H.maxabs(grads[k].cpu(), grads_f[k].cpu()), sc)
    g.close()


def test_graphed_training_step_follows_learning_rate_changes(AG, cuda):
    """a captured Adam must follow `update_learning_rate()` (base_model.py:131-138).  GraphedTrainStep feeds the
    replayed optimizer its learning rates through device scalars; after a scheduler step the replayed parameter update
    has to equal the eager one (and differ from a replay at the old rate)."""
    from eavsr_amd.eavsrp_model import EAVSRPModel
    from eavsr_amd.graph import GraphedTrainStep
    from eavsr_amd.utils.synthetic import synthetic_clip
    mk = lambda: Namespace(predict=False, n_frame=3, n_flow=5, scale=4, isTrain=True, gpu_ids=[0], lr=1e-4, beta1=0.9,
                           beta2=0.999, weight_decay=0.0, npost=350, lr_policy="step", lr_decay_iters=1, niter=10,
                           niter_decay=0, load_iter=0, load_path="", verbose=False)
    sd = H.filled(H.model_shapes("x4"), "trained_like")
    data = {"lr_seq": synthetic_clip(1, 3, 64, 64, seed=1), "hr_seq": synthetic_clip(1, 3, 256, 256, seed=2), "fname": "x"}
    batch = {k: v.to(cuda) for k, v in data.items() if k != "fname"}
    key = "backbone.backward_1.main.2.rg.0.res.0.weight"

    def run(graphed: bool, schedule: bool):
        m = EAVSRPModel(mk())
        m.netEAVSRP.load_state_dict(sd, strict=True)
        m.setup(m.opt)
        m.set_input(data, epoch=0)
        if graphed:
            g = GraphedTrainStep(m, warmup=1)         # one eager step, then the capture
            step = lambda: g.step(batch)
        else:
            m.optimize_parameters()
            step = m.optimize_parameters
        step()
        before = dict(m.netEAVSRP.named_parameters())[key].detach().clone()
        if schedule:
            m.update_learning_rate()                  # StepLR, step 1, gamma 0.5: both groups halve
            assert abs(m.optimizer_EAVSRP.param_groups[0]["lr"] - 5e-5) < 1e-12
        step()
        after = dict(m.netEAVSRP.named_parameters())[key].detach().clone()
        return (after - before).abs().mean().item()

    d_eager, d_graph, d_graph_const = run(False, True), run(True, True), run(True, False)
    assert abs(d_graph - d_eager) <= 0.05 * d_eager, (d_graph, d_eager)
    assert d_graph < 0.7 * d_graph_const, (d_graph, d_graph_const)     # the halved rate is what the replay applied


def test_propagate_partial_freeze_keeps_autograd(AG, cuda):
    """with the backbones and the encoder frozen but the alignment modules trained, `propagate` must not hand
    `upsample` a frame-major buffer whose rows the gradient-carrying sums never wrote: the output has to equal the
    all-trainable forward and the alignment parameters must receive gradients."""
    from eavsr_amd.eavsrp_model import EAVSRP
    from eavsr_amd.utils.synthetic import synthetic_clip
    opt = Namespace(predict=False, n_frame=3, n_flow=5, scale=4)
    sd = H.filled(H.model_shapes("x4"), "trained_like")
    clip, hr = synthetic_clip(1, 3, 64, 64, seed=5).to(cuda), synthetic_clip(1, 3, 256, 256, seed=6).to(cuda)

    def build(freeze):
        net = EAVSRP(opt, None)
        net.load_state_dict(sd, strict=True)
        net = net.to(cuda).train()
        if freeze:
            for name, p in net.named_parameters():
                if not name.startswith("deform_align."):
                    p.requires_grad_(False)
        return net
    full, part = build(False), build(True)
    y_full, y_part = full(clip), part(clip)
    assert torch.isfinite(y_part).all()
    assert H.maxabs(y_full.detach().cpu(), y_part.detach().cpu()) <= 1e-6
    (y_full - hr).abs().mean().backward()
    (y_part - hr).abs().mean().backward()
    gf, gp = dict(full.named_parameters()), dict(part.named_parameters())
    # the frozen part of `part` runs the inference kernels (Winograd), `full` the training kernels: equal to rounding.  Gradients
    # that are sums of cancelling terms (some biases: ~1e-9) are compared on the scale of the largest alignment gradient.
    gmax = max(gf[k].grad.abs().max().item() for k in gp if k.startswith("deform_align.") and gf[k].grad is not None)
    checked = 0
    for k, p in gp.items():
        if k.startswith("deform_align.") and gf[k].grad is not None:
            assert p.grad is not None, k
            scale = max(1e-8, gf[k].grad.abs().max().item())
            assert H.maxabs(p.grad.cpu(), gf[k].grad.cpu()) <= 5e-3 * scale + 1e-4 * gmax, k
            checked += 1
    assert checked > 50


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 96, 96), (1, 19, 37), (3, 40, 64), (2, 180, 320)])
def test_input_gradient_convolution_leaves_the_next_tail_backwards_plane_sums(cuda, shape):
    """conv2d(dY, W, dgrad=True, residual=d, sum_mul=m) (round 6; desc.sum_mul of eavsr_conv3x3_f32x6s): the same output bits as
    without sum_mul, and rows whose sum over the plane's tiles is sum_hw out * m -- in the epilogue of the small-launch bf16x6
    kernel at a training crop, by a plane-sum launch behind the kernels that take ragged widths / large launches"""
    from eavsr_amd import ops
    n, h, w = shape
    dy, d, m = cases.randn(41, n, 64, h, w).to(cuda), cases.randn(42, n, 64, h, w).to(cuda), cases.randn(43, n, 64, h, w).to(cuda)
    wt = cases.randn(44, 64, 64, 3, 3, scale=1.0 / 24).to(cuda)
    ref = ops.conv2d(dy, wt, None, residual=d, dgrad=True)
    with ops.profile() as prof:
        out, rows = ops.conv2d(dy, wt, None, residual=d, dgrad=True, sum_mul=m)
    assert torch.equal(out, ref)
    assert rows.dim() == 3 and rows.shape[0] == n and rows.shape[2] == 64
    want = (out.double() * m.double()).sum((2, 3))
    got = rows.double().sum(1)
    assert H.maxabs(got.cpu(), want.cpu()) <= 2e-6 * max(1.0, (out.double().abs() * m.double().abs()).sum((2, 3)).max().item())
    names = set(prof.summary())
    if shape == (2, 96, 96):
        assert names == {"conv3x3_64to64_x6s"}, names      # the epilogue form: ONE launch
    elif shape in ((1, 19, 37), (2, 180, 320)):
        assert "plane_sum" in names, names
    with pytest.raises(ValueError):
        ops.conv2d(dy, wt, None, sum_mul=m)      # forward convolutions do not take it


@pytest.mark.gpu
def test_consecutive_rcabs_hand_the_plane_sums_over(AG, cuda):
    """Four RCABs in a row (autograd._RcabFn): block k + 1's last input-gradient convolution leaves sum_hw dx * r_k in its epilogue,
    block k's tail backward adds the rows up itself -- one plane-sum launch (the LAST block's, whose d comes from outside) instead
    of four; gradients equal the unchained form's to rounding, and CPU autograd through the oracle's blocks; a block whose output
    is used twice (autograd ADDS the two gradients: the tensor it receives is not the one its successor returned) falls back"""
    from eavsr_amd import ops
    sd = H.filled(H.rcagroup_shapes("g.", 4), "trained_like")
    keys = [k for k in sd if k.startswith("g.rg.") and not k.startswith("g.rg.4.")]
    x, G = cases.randn(51, 2, 64, 96, 96), cases.randn(52, 2, 64, 96, 96)

    def run(chain, twice=False):
        AG.RCAB_CHAIN = chain
        gp = {k: torch.nn.Parameter(sd[k].to(cuda)) for k in keys}
        gx = leaf(x, cuda)
        with ops.profile() as prof:
            with AG.grad_sink():
                y, mid = gx * 1.0, None
                for b in range(4):
                    p = f"g.rg.{b}."
                    y = AG.rcab(y, gp[p + "res.0.weight"], gp[p + "res.0.bias"], gp[p + "res.2.weight"], gp[p + "res.2.bias"],
                                gp[p + "ca.conv_du.0.weight"], gp[p + "ca.conv_du.0.bias"], gp[p + "ca.conv_du.2.weight"],
                                gp[p + "ca.conv_du.2.bias"])
                    if b == 1:
                        mid = y
                loss = (y * G.to(cuda)).sum() + ((mid * mid).sum() * 0.5 if twice else 0.0)
                loss.backward()
        return [gx.grad] + [gp[k].grad for k in keys], prof.summary()

    try:
        got_c, prof_c = run(True)
        got_u, prof_u = run(False)
        got_t, prof_t = run(True, twice=True)
        got_tu, _ = run(False, twice=True)
    finally:
        AG.RCAB_CHAIN = True
    assert prof_u["plane_sum"]["calls"] == 4 and prof_c["plane_sum"]["calls"] == 1, (prof_u["plane_sum"], prof_c["plane_sum"])
    assert prof_t["plane_sum"]["calls"] == 2      # block 1's output has two consumers: its d is a sum autograd made
    for a, b, k in zip(got_c, got_u, ["x"] + keys):
        assert H.maxabs(a.cpu(), b.cpu()) <= 2e-6 * max(1.0, b.abs().max().item()), k
    for a, b, k in zip(got_t, got_tu, ["x"] + keys):
        assert H.maxabs(a.cpu(), b.cpu()) <= 2e-6 * max(1.0, b.abs().max().item()), k
    cp = {k: leaf(sd[k]) for k in keys}
    cx = leaf(x)
    y = cx
    for b in range(4):
        y = O.rcab({**sd, **cp}, f"g.rg.{b}.", y)
    ref = grads(y, G, [cx] + [cp[k] for k in keys])
    check(got_c, ref, 1e-4, ["x"] + keys)


def test_prepacked_weight_forms_equal_the_single_packs(cuda):
    """ops.prepack_conv3_x6 (eavsr_pack_conv_weight_x6_multi: up to 48 weights per launch, pointers by value): the forward and the
    input-gradient form of every weight equal the single-weight packs bit for bit, the cache serves them afterwards, a weight that
    changed in place (its version moved) is packed again and only that one"""
    from eavsr_amd import ops
    from eavsr_amd.graph import clear_weight_caches
    ws = [cases.randn(700 + i, 64, 64, 3, 3, scale=1.0 / 24).to(cuda) for i in range(50)]      # (> 48: two launches)
    clear_weight_caches()
    single = [(ops._packed_conv_x6([w]).clone(), ops._packed_conv_x6([w], dgrad=True).clone()) for w in ws]
    clear_weight_caches()
    assert ops.prepack_conv3_x6(ws + [cases.randn(9, 32, 64, 3, 3).to(cuda)]) == 100      # (another shape is left alone)
    for w, (f, d) in zip(ws, single):
        assert torch.equal(ops._packed_conv_x6([w]), f) and torch.equal(ops._packed_conv_x6([w], dgrad=True), d)
    assert ops.prepack_conv3_x6(ws) == 0
    ws[7].mul_(2.0)
    assert ops.prepack_conv3_x6(ws) == 2
    assert torch.equal(ops._packed_conv_x6([ws[7]]), ops._packed_conv_x6([ws[7].clone()]))
    x = cases.randn(3, 2, 64, 96, 96).to(cuda)
    y = ops.conv2d(x, ws[7], None)
    clear_weight_caches()
    assert torch.equal(ops.conv2d(x, ws[7], None), y)
