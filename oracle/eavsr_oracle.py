"""CPU oracle for the EAVSR alignment + propagation hot path.

TEST INFRASTRUCTURE ONLY.  This file is the *checker*: it may be imported by
``tests/``, by ``__graft_entry__.smoke()`` and by ``bench.py``'s ``cpu_baseline``
leg, and by nothing else.  The product (``eavsr_amd/``) never imports it and has
no CPU fallback.

It is a from-scratch, functional (state_dict in, tensors out) restatement in
plain fp32 PyTorch-CPU ops of the algorithm the reference implements in
``/root/reference/models/networks.py`` and ``/root/reference/models/eavsrp_model.py``.
Every function cites the reference lines it follows.  No reference source is
imported or copied here; parity with the reference is pinned by the golden
vectors under ``tests/golden/`` (made by ``tests/golden/gen_golden.py``, which
imports the real reference modules in the build container) and checked in
``tests/test_oracle_golden.py``.

PARITY STATUS
  * everything that lives in /root/reference (flow_warp, AdaptBlock2_3x3,
    AdaptBlockOffset, TransOffsetworelu, MultiAdSTN orchestration, CALayer,
    RCABlock, RCAGroup, ResidualBlocksWithInputConv, SPyNet, the encoder,
    propagate, upsample, EAVSRP.forward):   PINNED by golden vectors G1..G8.
  * DCNv2 (``mmcv.ops.modulated_deform_conv2d``, mmcv-full 1.x, version not
    pinned by the reference, source absent from /root/reference):
    PARITY UNPINNED against the mmcv binary.  ``dcnv2`` below restates the
    published mmcv 1.x algorithm (modulated_deform_conv_cuda_kernel.cuh:
    modulated_deformable_im2col + GEMM + bias) and is anchored on the reference's
    call site (networks.py:573,575-583,627-630), on how AdaptBlockOffset lays
    out offset/mask channels (networks.py:286-288,303-315), on a second
    independent restatement (``dcnv2_via_grid_sample``), on a C restatement
    (``oracle/dcnv2_ref.c``) and on known-answer identities (zero offset ==
    conv2d, integer offset == shifted conv2d, ...), see tests/test_oracle_dcn.py.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

BRANCHES = ("backward_1", "forward_1", "backward_2", "forward_2")  # eavsrp_model.py:141


# --------------------------------------------------------------------------------------
# a1 / a2  flow_warp
# --------------------------------------------------------------------------------------
def flow_warp_nhwc(x: Tensor, flow: Tensor, padding_mode: str = "zeros", interpolation: str = "bilinear",
                   align_corners: bool = True) -> Tensor:
    """eavsrp_model.py:587-626.  flow is (n,h,w,2), [...,0]=x displacement, [...,1]=y
    displacement, in pixels.  The path uses bilinear, align_corners=True; the other two arguments are forwarded to
    grid_sample exactly as the reference does (eavsrp_model.py:620-625)."""
    n, c, h, w = x.shape
    if tuple(flow.shape[1:3]) != (h, w):
        raise ValueError("The spatial sizes of input and flow are not the same.")
    gy, gx = torch.meshgrid(torch.arange(0, h), torch.arange(0, w), indexing="ij")
    grid = torch.stack((gx, gy), 2).to(x.dtype)  # (h,w,2)
    gf = grid + flow
    gfx = 2.0 * gf[..., 0] / max(w - 1, 1) - 1.0
    gfy = 2.0 * gf[..., 1] / max(h - 1, 1) - 1.0
    return F.grid_sample(x, torch.stack((gfx, gfy), dim=3), mode=interpolation,
                         padding_mode=padding_mode, align_corners=align_corners)


def flow_warp(x: Tensor, flow: Tensor, padding_mode: str = "zeros", interpolation: str = "bilinear",
              align_corners: bool = True) -> Tensor:
    """networks.py:699-739.  flow is NCHW (n,2,h,w): channel 0 = x, 1 = y displacement."""
    return flow_warp_nhwc(x, flow.permute(0, 2, 3, 1), padding_mode, interpolation, align_corners)


def flow_warp_direct(x: Tensor, flow: Tensor, padding_mode: str = "zeros") -> Tensor:
    """Second restatement of networks.py:699-739 that does not use grid_sample: explicit
    floor / 4-corner gather with the sampling coordinates computed directly in pixels.
    Used to cross-check ``flow_warp`` (differences are only float rounding of the
    normalise/un-normalise round trip the reference does)."""
    n, c, h, w = x.shape
    ys = torch.arange(h, dtype=x.dtype).view(1, h, 1)
    xs = torch.arange(w, dtype=x.dtype).view(1, 1, w)
    px = xs + flow[:, 0]
    py = ys + flow[:, 1]
    if padding_mode == "border":
        px = px.clamp(0, w - 1)
        py = py.clamp(0, h - 1)
    return _bilinear_zero(x, py, px)


def _bilinear_zero(x: Tensor, py: Tensor, px: Tensor) -> Tensor:
    """Corner-wise zero-padded bilinear sample of x (n,c,h,w) at pixel coords (n,ho,wo)."""
    n, c, h, w = x.shape
    y0 = torch.floor(py)
    x0 = torch.floor(px)
    ly = py - y0
    lx = px - x0
    out = x.new_zeros((n, c) + tuple(py.shape[1:]))
    flat = x.reshape(n, c, h * w)
    for dy, dx, wgt in ((0, 0, (1 - ly) * (1 - lx)), (0, 1, (1 - ly) * lx),
                        (1, 0, ly * (1 - lx)), (1, 1, ly * lx)):
        yy = y0 + dy
        xx = x0 + dx
        ok = (yy >= 0) & (yy <= h - 1) & (xx >= 0) & (xx <= w - 1)
        idx = (yy.clamp(0, h - 1) * w + xx.clamp(0, w - 1)).long()
        v = torch.gather(flat, 2, idx.reshape(n, 1, -1).expand(n, c, -1)).reshape(out.shape)
        out = out + v * (wgt * ok.to(x.dtype)).unsqueeze(1)
    return out


# --------------------------------------------------------------------------------------
# a7  DCNv2  (third party: mmcv 1.x; see module docstring -- parity unpinned)
# --------------------------------------------------------------------------------------
def dcnv2(x: Tensor, offset: Tensor, mask: Tensor, weight: Tensor, bias: Tensor | None,
          stride: int = 1, padding: int = 1, dilation: int = 1, groups: int = 1,
          deform_groups: int = 1) -> Tensor:
    """Modulated deformable convolution as consumed at networks.py:627-630.

    offset channel order  g*2K + 2k + {0: dy, 1: dx};  mask order g*K + k;  tap k=i*kw+j
    samples at (y*stride - pad + i*dil + dy, x*stride - pad + j*dil + dx); the sample is
    taken only if -1 < p < size in both dims and is corner-wise zero-padded bilinear;
    col[c*K+k] = sample * mask[g(c), k];  out = W(co, c*K+k) . col + b.
    """
    n, c, h, w = x.shape
    co, cig, kh, kw = weight.shape
    if groups != 1:
        raise NotImplementedError("the reference only uses groups=1 (networks.py:577-583)")
    K = kh * kw
    dg = deform_groups
    ho = (h + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    wo = (w + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    assert offset.shape == (n, dg * 2 * K, ho, wo), offset.shape
    assert mask.shape == (n, dg * K, ho, wo), mask.shape
    cpg = c // dg
    ys = (torch.arange(ho, dtype=x.dtype) * stride - padding).view(1, ho, 1)
    xs = (torch.arange(wo, dtype=x.dtype) * stride - padding).view(1, 1, wo)
    cols = x.new_zeros(n, c, K, ho, wo)
    for g in range(dg):
        xg = x[:, g * cpg:(g + 1) * cpg]
        for i in range(kh):
            for j in range(kw):
                k = i * kw + j
                py = ys + i * dilation + offset[:, g * 2 * K + 2 * k]
                px = xs + j * dilation + offset[:, g * 2 * K + 2 * k + 1]
                valid = ((py > -1) & (px > -1) & (py < h) & (px < w)).to(x.dtype)
                v = _bilinear_zero(xg, py, px) * valid.unsqueeze(1)
                cols[:, g * cpg:(g + 1) * cpg, k] = v * mask[:, g * K + k].unsqueeze(1)
    out = torch.einsum("ok,nkp->nop", weight.reshape(co, c * K), cols.reshape(n, c * K, ho * wo))
    out = out.reshape(n, co, ho, wo)
    if bias is not None:
        out = out + bias.view(1, co, 1, 1)
    return out


def dcnv2_via_grid_sample(x, offset, mask, weight, bias, stride=1, padding=1, dilation=1,
                          groups=1, deform_groups=1):
    """Independent second restatement: the same sampling positions fed to
    F.grid_sample(padding_mode='zeros', align_corners=True) on pixel coordinates."""
    n, c, h, w = x.shape
    co, _, kh, kw = weight.shape
    K = kh * kw
    dg = deform_groups
    cpg = c // dg
    ho, wo = offset.shape[-2:]
    ys = (torch.arange(ho, dtype=x.dtype) * stride - padding).view(1, ho, 1)
    xs = (torch.arange(wo, dtype=x.dtype) * stride - padding).view(1, 1, wo)
    cols = []
    for cc in range(c):
        g = cc // cpg
        for i in range(kh):
            for j in range(kw):
                k = i * kw + j
                py = ys + i * dilation + offset[:, g * 2 * K + 2 * k]
                px = xs + j * dilation + offset[:, g * 2 * K + 2 * k + 1]
                grid = torch.stack((2 * px / max(w - 1, 1) - 1, 2 * py / max(h - 1, 1) - 1), -1)
                v = F.grid_sample(x[:, cc:cc + 1], grid, mode="bilinear", padding_mode="zeros",
                                  align_corners=True)[:, 0]
                cols.append(v * mask[:, g * K + k])
    cols = torch.stack(cols, 1).reshape(n, c * K, ho * wo)
    out = torch.einsum("ok,nkp->nop", weight.reshape(co, c * K), cols).reshape(n, co, ho, wo)
    if bias is not None:
        out = out + bias.view(1, co, 1, 1)
    return out


# --------------------------------------------------------------------------------------
# a3 / a4 / a6   offset predictors
# --------------------------------------------------------------------------------------
_REGULAR = torch.tensor([[-1, -1, -1, 0, 0, 0, 1, 1, 1],
                         [-1, 0, 1, -1, 0, 1, -1, 0, 1]], dtype=torch.float32)  # networks.py:286-288


def adapt_frontend(sd: SD, p: str, x: Tensor, h_hr: Tensor) -> Tensor:
    """networks.py:290-291,300 (and :327-328,336): cat -> depthwise 3x3 + LeakyReLU(0.2)
    -> grouped 3x3 (groups = C, 2 in-channels per out-channel) + LeakyReLU(0.2)."""
    c = x.shape[1]
    t = torch.cat([x, h_hr], dim=1)
    t = F.leaky_relu(F.conv2d(t, sd[p + "concat.0.weight"], sd[p + "concat.0.bias"], 1, 1, 1, 2 * c), 0.2)
    t = F.leaky_relu(F.conv2d(t, sd[p + "concat2.0.weight"], sd[p + "concat2.0.bias"], 1, 1, 1, c), 0.2)
    return t


def affine_offsets(transform: Tensor, translation: Tensor, D: int) -> Tensor:
    """networks.py:302-311 (D groups) / :338-346 (D=1):  per pixel and group
    off(2x9) = T(2x2) @ R(2x9) - R, laid out as channels g*18 + 2k + {0:y,1:x}, then
    += translation (channel 0 -> y/even, 1 -> x/odd)."""
    n, _, h, w = transform.shape
    T = transform.reshape(n, D, 2, 2, h, w)
    R = _REGULAR.to(transform.dtype)
    offy = T[:, :, 0, 0, None] * R[0].view(1, 1, 9, 1, 1) + T[:, :, 0, 1, None] * R[1].view(1, 1, 9, 1, 1) \
        - R[0].view(1, 1, 9, 1, 1)
    offx = T[:, :, 1, 0, None] * R[0].view(1, 1, 9, 1, 1) + T[:, :, 1, 1, None] * R[1].view(1, 1, 9, 1, 1) \
        - R[1].view(1, 1, 9, 1, 1)
    tr = translation.reshape(n, D, 2, 1, h, w)
    offy = offy + tr[:, :, 0]
    offx = offx + tr[:, :, 1]
    return torch.stack((offy, offx), dim=3).reshape(n, D * 18, h, w)  # (n,D,9,2,h,w)


def adapt_block2_3x3(sd: SD, p: str, x: Tensor, h_hr: Tensor) -> Tensor:
    """AdaptBlock2_3x3.forward networks.py:334-348 -> (n,18,h,w)."""
    f = adapt_frontend(sd, p, x, h_hr)
    tm = F.conv2d(f, sd[p + "transform_matrix_conv.weight"], sd[p + "transform_matrix_conv.bias"], 1, 1)
    tl = F.conv2d(f, sd[p + "translation_conv.weight"], sd[p + "translation_conv.bias"], 1, 1)
    return affine_offsets(tm, tl, 1)


def trans_offset(sd: SD, p: str, off: Tensor) -> Tensor:
    """TransOffsetworelu.forward networks.py:566-571: 3x3 conv 18 -> 2, no activation."""
    return F.conv2d(off, sd[p + "conv_first.weight"], sd[p + "conv_first.bias"], 1, 1)


def adapt_block_offset(sd: SD, p: str, x: Tensor, h_hr: Tensor, D: int = 8):
    """AdaptBlockOffset.forward networks.py:298-315 -> (offset (n,18D,h,w), mask (n,9D,h,w))."""
    f = adapt_frontend(sd, p, x, h_hr)
    tm = F.conv2d(f, sd[p + "transform_matrix_conv.weight"], sd[p + "transform_matrix_conv.bias"], 1, 2)
    tl = F.conv2d(f, sd[p + "translation_conv.weight"], sd[p + "translation_conv.bias"], 1, 2)
    mk = torch.sigmoid(F.conv2d(f, sd[p + "mask_conv.weight"], sd[p + "mask_conv.bias"], 1, 2))
    return affine_offsets(tm, tl, D), mk


# --------------------------------------------------------------------------------------
# a5  MultiAdSTN
# --------------------------------------------------------------------------------------
def _interp_ac(x: Tensor, s: float) -> Tensor:
    return F.interpolate(x, scale_factor=s, mode="bilinear", align_corners=True)


def multi_adstn(sd: SD, p: str, nbr_feat_l: Sequence[Tensor], ref_feat_l: Sequence[Tensor],
                feat_prop: Tensor, offset: Tensor, D: int = 8, return_parts: bool = False):
    """MultiAdSTN.forward networks.py:597-631 (flag=False)."""
    off_d4 = _interp_ac(offset, 0.25) / 4.0
    off_d2 = _interp_ac(offset, 0.5) / 2.0
    # level 3 (h/4)  :604-608
    w4 = flow_warp(nbr_feat_l[2], off_d4)
    p1 = trans_offset(sd, p + "trans_l3.", adapt_block2_3x3(sd, p + "flow_l3.", w4, ref_feat_l[2]))
    p1_up = _interp_ac(p1, 2) * 2
    # level 2 (h/2)  :609-613
    w2 = flow_warp(nbr_feat_l[1], off_d2 + p1_up)
    p2 = trans_offset(sd, p + "trans_l2.", adapt_block2_3x3(sd, p + "flow_l2.", w2, ref_feat_l[1]))
    p2_up = _interp_ac(p2 + p1_up, 2) * 2
    # level 1 (h)    :614-619
    w1 = flow_warp(nbr_feat_l[0], offset + p2_up)
    p3 = trans_offset(sd, p + "trans_l1.", adapt_block2_3x3(sd, p + "flow_l1.", w1, ref_feat_l[0]))
    offset = p3 + p2_up + offset
    nbr = flow_warp(nbr_feat_l[0], offset)          # :621
    feat = flow_warp(feat_prop, offset)             # :623
    de_offset, mask = adapt_block_offset(sd, p + "adastn.", nbr, ref_feat_l[0], D)  # :625
    out = dcnv2(feat, de_offset, mask, sd[p + "weight"], sd[p + "bias"], 1, 1, 1, 1, D)  # :627-630
    if return_parts:
        return out, dict(offset=offset, de_offset=de_offset, mask=mask, feat=feat, nbr=nbr)
    return out


# --------------------------------------------------------------------------------------
# a10 / a11   residual backbone
# --------------------------------------------------------------------------------------
def ca_layer(sd: SD, p: str, x: Tensor) -> Tensor:
    """CALayer.forward networks.py:444-447."""
    y = x.mean(dim=(2, 3), keepdim=True)
    y = F.relu(F.conv2d(y, sd[p + "conv_du.0.weight"], sd[p + "conv_du.0.bias"]))
    y = torch.sigmoid(F.conv2d(y, sd[p + "conv_du.2.weight"], sd[p + "conv_du.2.bias"]))
    return x * y


def rcab(sd: SD, p: str, x: Tensor) -> Tensor:
    """RCABlock.forward networks.py:461-464, mode 'CRC'."""
    r = F.relu(F.conv2d(x, sd[p + "res.0.weight"], sd[p + "res.0.bias"], 1, 1))
    r = F.conv2d(r, sd[p + "res.2.weight"], sd[p + "res.2.bias"], 1, 1)
    return ca_layer(sd, p + "ca.", r) + x


def rca_group(sd: SD, p: str, x: Tensor, nb: int) -> Tensor:
    """RCAGroup.forward networks.py:480-482: nb RCABs, conv, + x."""
    r = x
    for k in range(nb):
        r = rcab(sd, f"{p}rg.{k}.", r)
    r = F.conv2d(r, sd[f"{p}rg.{nb}.weight"], sd[f"{p}rg.{nb}.bias"], 1, 1)
    return r + x


def count_blocks(sd: SD, p: str) -> int:
    nb = 0
    while f"{p}main.2.rg.{nb}.res.0.weight" in sd:
        nb += 1
    return nb


def resblocks_with_input_conv(sd: SD, p: str, feat: Tensor) -> Tensor:
    """ResidualBlocksWithInputConv.forward eavsrp_model.py:375-400."""
    x = F.leaky_relu(F.conv2d(feat, sd[p + "main.0.weight"], sd[p + "main.0.bias"], 1, 1), 0.1)
    return rca_group(sd, p + "main.2.", x, count_blocks(sd, p))


# --------------------------------------------------------------------------------------
# callers either side of the path: SPyNet (f1), encoder / upsample (f2)
# --------------------------------------------------------------------------------------
_MEAN = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
_STD = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)


def spynet_basic(sd: SD, p: str, t: Tensor) -> Tensor:
    """SPyNetBasicModule eavsrp_model.py:534-585: 5 x (7x7 conv), ReLU on the first 4."""
    for i in range(5):
        t = F.conv2d(t, sd[f"{p}basic_module.{i}.conv.weight"], sd[f"{p}basic_module.{i}.conv.bias"], 1, 3)
        if i < 4:
            t = F.relu(t)
    return t


def spynet_compute_flow(sd: SD, p: str, ref: Tensor, supp: Tensor) -> Tensor:
    """SPyNet.compute_flow eavsrp_model.py:433-488."""
    n, _, h, w = ref.shape
    mean, std = sd.get(p + "mean", _MEAN), sd.get(p + "std", _STD)
    refs = [(ref - mean) / std]
    supps = [(supp - mean) / std]
    for _ in range(5):
        refs.append(F.avg_pool2d(refs[-1], 2, 2, count_include_pad=False))
        supps.append(F.avg_pool2d(supps[-1], 2, 2, count_include_pad=False))
    refs, supps = refs[::-1], supps[::-1]
    flow = ref.new_zeros(n, 2, h // 32, w // 32)
    for level in range(6):
        flow_up = flow if level == 0 else _interp_ac(flow, 2) * 2.0
        warped = flow_warp_nhwc(supps[level], flow_up.permute(0, 2, 3, 1), "border")
        flow = flow_up + spynet_basic(sd, f"{p}basic_module.{level}.",
                                      torch.cat([refs[level], warped, flow_up], 1))
    return flow


def spynet(sd: SD, p: str, ref: Tensor, supp: Tensor) -> Tensor:
    """SPyNet.forward eavsrp_model.py:490-523."""
    h, w = ref.shape[2:4]
    w_up = w if w % 32 == 0 else 32 * (w // 32 + 1)
    h_up = h if h % 32 == 0 else 32 * (h // 32 + 1)
    ref = F.interpolate(ref, size=(h_up, w_up), mode="bilinear", align_corners=False)
    supp = F.interpolate(supp, size=(h_up, w_up), mode="bilinear", align_corners=False)
    flow = F.interpolate(spynet_compute_flow(sd, p, ref, supp), size=(h, w), mode="bilinear",
                         align_corners=False)
    flow = flow.clone()
    flow[:, 0] *= float(w) / float(w_up)
    flow[:, 1] *= float(h) / float(h_up)
    return flow


def compute_flow(sd: SD, lrs: Tensor):
    """EAVSRP.compute_flow eavsrp_model.py:179-200."""
    n, t, c, h, w = lrs.shape
    l1 = lrs[:, :-1].reshape(-1, c, h, w)
    l2 = lrs[:, 1:].reshape(-1, c, h, w)
    fb = spynet(sd, "spynet.", l1, l2).view(n, t - 1, 2, h, w)
    ff = spynet(sd, "spynet.", l2, l1).view(n, t - 1, 2, h, w)
    return ff, fb


_VGG = ("conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1")


def encoder(sd: SD, p: str, x: Tensor) -> Tensor:
    """ContrasExtractorLayer.forward networks.py:549-552: normalise, VGG16 conv1_1..conv3_1
    with the two pools removed (ReLU after each but the last, which is features[:11]'s end),
    then tail conv 256->64."""
    x = (x - sd.get(p + "mean", _MEAN)) / sd.get(p + "std", _STD)
    for i, name in enumerate(_VGG):
        x = F.conv2d(x, sd[f"{p}model.{name}.weight"], sd[f"{p}model.{name}.bias"], 1, 1)
        if i < len(_VGG) - 1:
            x = F.relu(x)
    return F.conv2d(x, sd[p + "tail.weight"], sd[p + "tail.bias"], 1, 1)


def propagate(sd: SD, feats: Dict[str, List[Tensor]], flows: Tensor, module_name: str,
              D: int = 8) -> Dict[str, List[Tensor]]:
    """EAVSRP.propagate eavsrp_model.py:242-329."""
    n, t, _, h, w = flows.shape
    frame_idx = list(range(0, t + 1))
    flow_idx = list(range(-1, t))
    mapping_idx = list(range(0, len(feats["spatial"])))
    mapping_idx += mapping_idx[::-1]
    backward = "backward" in module_name
    if backward:
        frame_idx = frame_idx[::-1]
        flow_idx = frame_idx
    feat_prop = flows.new_zeros(n, 64, h, w)
    pa = f"deform_align.{module_name}."
    step = 1 if backward else -1
    for i, idx in enumerate(frame_idx):
        cur = [feats[k][mapping_idx[idx]] for k in ("spatial", "spatial_d2", "spatial_d4")]
        if i > 0:
            nbr = [feats[k][mapping_idx[idx + step]] for k in ("spatial", "spatial_d2", "spatial_d4")]
            flow_n1 = flows[:, flow_idx[i]]
            cond_n1 = multi_adstn(sd, pa, nbr, cur, feat_prop, flow_n1, D)
            cond_n2 = torch.zeros_like(cond_n1)
            if i > 1:
                feat_n2 = feats[module_name][-2]
                nbr2 = [feats[k][mapping_idx[idx + 2 * step]] for k in ("spatial", "spatial_d2", "spatial_d4")]
                flow_n2 = flows[:, flow_idx[i - 1]]
                flow_n2 = flow_n1 + flow_warp_nhwc(flow_n2, flow_n1.permute(0, 2, 3, 1))
                cond_n2 = multi_adstn(sd, pa, nbr2, cur, feat_n2, flow_n2, D)
            feat_prop = torch.cat([cond_n1, cur[0], cond_n2], dim=1)
            feat_prop = F.conv2d(feat_prop, sd[f"fusion.{module_name}.weight"], sd[f"fusion.{module_name}.bias"])
        feat = [cur[0]] + [feats[k][idx] for k in feats
                           if k not in ("spatial", "spatial_d2", "spatial_d4", module_name)] + [feat_prop]
        feat_prop = feat_prop + resblocks_with_input_conv(sd, f"backbone.{module_name}.", torch.cat(feat, dim=1))
        feats[module_name].append(feat_prop)
    if backward:
        feats[module_name] = feats[module_name][::-1]
    return feats


def upsample(sd: SD, lqs: Tensor, feats: Dict[str, List[Tensor]], scale: int = 4) -> Tensor:
    """EAVSRP.upsample eavsrp_model.py:331-364 (x2 twin: eavsrpx2_model.py:334-365, one
    pixel-shuffle stage and a x2 bilinear skip)."""
    outs = []
    t = lqs.shape[1]
    for i in range(t):
        hr = [feats["spatial"][i]] + [feats[k][i] for k in feats
                                      if k not in ("spatial", "spatial_d2", "spatial_d4")]
        hr = resblocks_with_input_conv(sd, "reconstruction.", torch.cat(hr, dim=1))
        hr = F.leaky_relu(F.pixel_shuffle(F.conv2d(hr, sd["upsample1.0.weight"], sd["upsample1.0.bias"], 1, 1), 2), 0.1)
        if scale == 4:
            hr = F.leaky_relu(F.pixel_shuffle(F.conv2d(hr, sd["upsample2.0.weight"], sd["upsample2.0.bias"], 1, 1), 2), 0.1)
        hr = F.leaky_relu(F.conv2d(hr, sd["conv_hr.weight"], sd["conv_hr.bias"], 1, 1), 0.1)
        hr = F.conv2d(hr, sd["conv_last.weight"], sd["conv_last.bias"], 1, 1)
        hr = hr + F.interpolate(lqs[:, i], scale_factor=scale, mode="bilinear", align_corners=False)
        outs.append(hr)
    return torch.stack(outs, dim=1)


def feature_pyramid(lr_feature: Tensor):
    """eavsrp_model.py:218-220 (a12)."""
    d2 = F.interpolate(lr_feature, scale_factor=0.5, mode="bilinear", align_corners=False)
    d4 = F.interpolate(lr_feature, scale_factor=0.25, mode="bilinear", align_corners=False)
    return d2, d4


def eavsrp_forward(sd: SD, lrs: Tensor, scale: int = 4, D: int = 8, flows=None, return_feats: bool = False):
    """EAVSRP.forward eavsrp_model.py:202-240: (n,t,3,h,w) -> (n,t,3,s*h,s*w).  return_feats: also the propagated branch
    features (the `feats` dict that eavsrp_model.py:229-240 hands to `upsample`), for tests that compare them directly."""
    n, t, c, h, w = lrs.shape
    assert h >= 64 and w >= 64
    with torch.no_grad():
        ff, fb = compute_flow(sd, lrs) if flows is None else flows
        f1 = encoder(sd, "encoder.", lrs.reshape(-1, c, h, w))
        d2, d4 = feature_pyramid(f1)
        f1 = f1.view(n, t, -1, h, w)
        d2 = d2.view(n, t, -1, h // 2, w // 2)
        d4 = d4.view(n, t, -1, h // 4, w // 4)
        feats = {"spatial": [f1[:, i] for i in range(t)],
                 "spatial_d2": [d2[:, i] for i in range(t)],
                 "spatial_d4": [d4[:, i] for i in range(t)]}
        for it in (1, 2):
            for direction in ("backward", "forward"):
                module = f"{direction}_{it}"
                feats[module] = []
                feats = propagate(sd, feats, fb if direction == "backward" else ff, module, D)
        out = upsample(sd, lrs, feats, scale)
        return (out, feats) if return_feats else out


# --------------------------------------------------------------------------------------
# metric helper (util/util.py:302-320, base_model.py:145-150)
# --------------------------------------------------------------------------------------
def psnr_255(sr: Tensor, hr: Tensor) -> float:
    a = (sr.clamp(0, 1) * 255).round()
    b = (hr.clamp(0, 1) * 255).round()
    mse = (((a - b) / 255.0) ** 2).mean().item()
    return float("inf") if mse == 0 else -10 * math.log10(mse)
