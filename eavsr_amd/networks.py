"""Drop-in counterparts of the hot-path classes of the reference's models/networks.py.

Same class names, constructor arguments, forward() signatures and state_dict keys as the
reference (so `load_networks`, which exits on any key mismatch -- base_model.py:193-213 -- keeps
working), but every forward() runs hand-written HIP kernels from libeavsr_hip.so through
eavsr_amd.ops.  The modules can be *constructed* and (de)serialised on any device; calling them
needs fp32 tensors on a MI355X -- there is no CPU or PyTorch-op fallback.

Only what EAVSRP instantiates is here (rows 2-6, 8); the dead classes of
networks.py (PatchSelect, ResBlock*, Flownet, SPYAdaSTN, ...) are out of scope.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import List, Sequence, Union

import torch
import torch.nn as nn

from . import autograd as AG
from . import ops

Tensor = torch.Tensor
TensorOrList = Union[Tensor, Sequence[Tensor]]


# --------------------------------------------------------------------------------------------
# layer-string builder (networks.py:84-95 `seq`, :108-190 `conv`): only the modes the hot path uses
# --------------------------------------------------------------------------------------------
def seq(*args):
    if len(args) == 1:
        args = args[0]
    if isinstance(args, nn.Module):
        return args
    return nn.Sequential(*[seq(a) for a in args])


class Conv2d(nn.Conv2d):
    """nn.Conv2d parameter container (same keys / default init) whose forward is the MFMA conv.
    Dense stride-1 "same" convolutions only; `act` fuses the activation that follows it in the
    reference's nn.Sequential."""

    def forward(self, x: TensorOrList, act=None, slope=0.0, residual=None, chan_partial=False, pixel_shuffle2=False):
        if self.groups != 1 or self.stride != (1, 1) or self.dilation != (1, 1) or \
                self.padding != (self.kernel_size[0] // 2,) * 2 or self.padding_mode != "zeros":
            raise NotImplementedError("eavsr_amd Conv2d: dense stride-1 same-padding convolutions only")
        return AG.conv2d(x, self.weight, self.bias, act=act, slope=slope, residual=residual,
                         chan_partial=chan_partial, pixel_shuffle2=pixel_shuffle2)


class _Act(nn.Module):
    """Placeholder that keeps nn.Sequential indices (hence state_dict keys) identical to the
    reference; the activation itself is fused into the preceding conv's epilogue."""

    def __init__(self, kind: str, slope: float = 0.0):
        super().__init__()
        self.kind, self.slope = kind, slope

    def extra_repr(self):
        return f"{self.kind}, slope={self.slope} (fused)"


def conv(in_channels=64, out_channels=64, kernel_size=3, stride=1, padding=1, bias=True, groups=1, mode="CBR"):
    """networks.py:108-190 for the modes EAVSRP uses: 'C', 'R', 'L' (LeakyReLU 0.2)."""
    L = []
    for t in mode:
        if t == "C":
            L.append(Conv2d(in_channels, out_channels, kernel_size, stride, padding, groups=groups, bias=bias))
        elif t in "Rr":
            L.append(_Act("relu"))
        elif t in "Ll":
            L.append(_Act("lrelu", 0.2))
        else:
            raise NotImplementedError(f"conv mode {t!r} is not used on the EAVSR hot path")
    return seq(*L)


def _run_fused(seq_mod: nn.Sequential, x: TensorOrList, residual=None, chan_partial=False):
    """Run a Sequential of Conv2d / _Act with each activation fused into the conv before it."""
    mods = list(seq_mod) if isinstance(seq_mod, nn.Sequential) else [seq_mod]
    i, out = 0, x
    while i < len(mods):
        m = mods[i]
        if not isinstance(m, Conv2d):
            raise NotImplementedError(type(m))
        act, slope = None, 0.0
        if i + 1 < len(mods) and isinstance(mods[i + 1], _Act):
            act, slope = mods[i + 1].kind, mods[i + 1].slope
            i += 1
        last = i == len(mods) - 1
        out = m(out, act=act, slope=slope, residual=residual if last else None,
                chan_partial=chan_partial and last)
        i += 1
    return out


# --------------------------------------------------------------------------------------------
# flow_warp  (networks.py:699-739)
# --------------------------------------------------------------------------------------------
def flow_warp(x, flow, interpolation="bilinear", padding_mode="zeros", align_corners=True):
    """Warp `x` (n,c,h,w) by `flow` (n,2,h,w; channel 0 = x displacement, 1 = y, in pixels)."""
    return AG.flow_warp(x, flow, padding_mode=padding_mode, flow_layout="nchw", interpolation=interpolation,
                        align_corners=align_corners)


# --------------------------------------------------------------------------------------------
# DCNv2 (mmcv.ops stand-ins: networks.py:573)
# --------------------------------------------------------------------------------------------
modulated_deform_conv2d = AG.modulated_deform_conv2d


class ModulatedDeformConv2d(nn.Module):
    """Parameter-compatible with mmcv.ops.ModulatedDeformConv2d (weight, bias; mmcv's default init
    uniform(+-1/sqrt(cin*kh*kw)), zero bias)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deform_groups=1, bias=True):
        super().__init__()
        ks = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, ks
        self.stride = (stride, stride) if isinstance(stride, int) else tuple(stride)
        self.padding = (padding, padding) if isinstance(padding, int) else tuple(padding)
        self.dilation = (dilation, dilation) if isinstance(dilation, int) else tuple(dilation)
        self.groups, self.deform_groups = groups, deform_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *ks))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.init_weights()

    def init_weights(self):
        n = self.in_channels
        for k in self.kernel_size:
            n *= k
        stdv = 1.0 / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.zero_()

    def forward(self, x, offset, mask):
        return modulated_deform_conv2d(x, offset, mask, self.weight, self.bias, self.stride, self.padding,
                                       self.dilation, self.groups, self.deform_groups)


# --------------------------------------------------------------------------------------------
# offset predictors (networks.py:280-348, 566-571)
# --------------------------------------------------------------------------------------------
_REGULAR = [[-1, -1, -1, 0, 0, 0, 1, 1, 1], [-1, 0, 1, -1, 0, 1, -1, 0, 1]]


class _AdaptBase(nn.Module):
    def __init__(self, inplanes):
        super().__init__()
        self.register_buffer("regular_matrix", torch.tensor(_REGULAR).float())
        self.concat = conv(inplanes * 2, inplanes * 2, groups=inplanes * 2, mode="CL")
        self.concat2 = conv(inplanes * 2, inplanes, groups=inplanes, mode="CL")

    def _frontend(self, x, h_hr):
        c1, c2 = self.concat[0], self.concat2[0]
        return AG.adapt_frontend(x, h_hr, c1.weight, c1.bias, c2.weight, c2.bias)


class AdaptBlock2_3x3(_AdaptBase):
    """networks.py:318-348: (x, h_hr) -> 18 sampling offsets of one 3x3 grid."""

    def __init__(self, opt, inplanes=64, outplanes=64, stride=1, dilation=1, deformable_groups=64):
        super().__init__(inplanes)
        self.opt = opt
        self.mask = True
        self.transform_matrix_conv = Conv2d(inplanes, 4, 3, 1, 1, bias=True)
        self.translation_conv = Conv2d(inplanes, 2, 3, 1, 1, bias=True)

    def forward(self, x, h_hr):
        f = self._frontend(x, h_hr)
        heads = AG.conv2d(f, [self.transform_matrix_conv.weight, self.translation_conv.weight],
                          [self.transform_matrix_conv.bias, self.translation_conv.bias])
        return AG.affine_offsets(heads, 1, with_mask=False)[0]


class AdaptBlockOffset(_AdaptBase):
    """networks.py:280-315: (x, h_hr) -> (offset (n,18D,h,w) in mmcv order, mask (n,9D,h,w))."""

    def __init__(self, opt, inplanes=64, outplanes=64, stride=1, dilation=1, deformable_groups=64):
        super().__init__(inplanes)
        self.D = deformable_groups
        self.opt = opt
        self.transform_matrix_conv = Conv2d(inplanes, 4 * self.D, 5, 1, 2, bias=True)
        self.translation_conv = Conv2d(inplanes, 2 * self.D, 5, 1, 2, bias=True)
        self.mask_conv = Conv2d(inplanes, 9 * self.D, 5, 1, 2, bias=True)
        self.relu = _Act("lrelu", 0.2)  # unused by forward in the reference as well

    def heads(self, x, h_hr, mask_activated=False):
        """the three 5x5 heads as one (n, 15 D, h, w) tensor: transform g*4+{0..3}, translation 4D + g*2+{0,1}, mask logits
        6D + g*9+k -- what `forward` expands into (offset, mask) and what the fused DCNv2 kernel consumes directly.
        mask_activated: the 9 D mask channels leave as sigmoid(logit) (networks.py:313-314 applied in the convolution's epilogue)"""
        f = self._frontend(x, h_hr)
        ws = [self.transform_matrix_conv.weight, self.translation_conv.weight, self.mask_conv.weight]
        bs = [self.transform_matrix_conv.bias, self.translation_conv.bias, self.mask_conv.bias]
        if (BACKBONE_DTYPE is not None and HEADS_IN_16BIT and f.shape[1] == 64 and 15 * self.D <= 128
                and not AG.needs_grad(f, ws, bs)):
            # 16-bit modes (BASELINE configs[2] / [4]): operands rounded once to bf16 / fp16, fp32 accumulation, fp32 heads out
            return ops.conv5x5_c64_h16(ops.to_nhwc_h16(f, BACKBONE_DTYPE), ws, bs)
        if mask_activated:
            return ops.conv2d(f, ws, bs, sigmoid_from=6 * self.D)
        return AG.conv2d(f, ws, bs)

    def forward(self, x, h_hr):
        return AG.affine_offsets(self.heads(x, h_hr), self.D, with_mask=True)


class TransOffsetworelu(nn.Module):
    """networks.py:566-571: 3x3 conv 18 -> 2 ("offsets -> residual flow"), no activation."""

    def __init__(self):
        super().__init__()
        self.conv_first = conv(18, 2, mode="C")

    def forward(self, offset):
        return self.conv_first(offset)


class MultiAdSTN(ModulatedDeformConv2d):
    """networks.py:575-631.  forward(nbr_feat_l, ref_feat_l, feat_prop, offset, flag=False)."""

    def __init__(self, opt, inplanes=64, outplanes=64, stride=1, dilation=1, deformable_groups=64):
        super().__init__(inplanes, outplanes, kernel_size=3, padding=1, stride=stride, dilation=dilation,
                         deform_groups=deformable_groups)
        self.opt = opt
        kw = dict(inplanes=inplanes, outplanes=outplanes, stride=stride, dilation=dilation,
                  deformable_groups=deformable_groups)
        self.flow_l1 = AdaptBlock2_3x3(opt, **kw)
        self.flow_l2 = AdaptBlock2_3x3(opt, **kw)
        self.flow_l3 = AdaptBlock2_3x3(opt, **kw)
        self.adastn = AdaptBlockOffset(opt, **kw)
        self.trans_l3 = TransOffsetworelu()
        self.trans_l2 = TransOffsetworelu()
        self.trans_l1 = TransOffsetworelu()
        self.center = getattr(opt, "n_frame", 7) // 2

    @staticmethod
    def _level(flow_blk, trans_blk, warp, ref):
        """trans(flow(warp, ref)): one pyramid level's residual flow (networks.py:605-607 etc.).  Inference: ONE kernel
        (front end, 64 -> 4 + 2 heads, affine -> 18 offsets, 18 -> 2 conv, nothing but the two inputs and the 2-channel flow
        touch HBM); with gradients: the un-fused ops, each with its backward kernel."""
        tconv = trans_blk.conv_first[0] if isinstance(trans_blk.conv_first, nn.Sequential) else trans_blk.conv_first
        params = [warp, ref] + list(flow_blk.parameters()) + list(trans_blk.parameters())
        if FUSE_FLOW_LEVEL and not AG.needs_grad(params) and warp.shape[1] % 2 == 0:
            c1, c2 = flow_blk.concat[0], flow_blk.concat2[0]
            return ops.flow_level(warp, ref, c1.weight, c1.bias, c2.weight, c2.bias,
                                  [flow_blk.transform_matrix_conv.weight, flow_blk.translation_conv.weight],
                                  [flow_blk.transform_matrix_conv.bias, flow_blk.translation_conv.bias],
                                  tconv.weight, tconv.bias)
        return trans_blk(flow_blk(warp, ref))

    def _fused_alignment(self, nbr, feat_prop, offset) -> bool:
        return (ops.DCN_MODE in ("il6", "il9") and not AG.needs_grad(nbr, feat_prop, offset, list(self.parameters()))
                and self.kernel_size == (3, 3) and self.stride == (1, 1) and self.padding == (1, 1)
                and self.dilation == (1, 1) and self.groups == 1 and self.in_channels % 8 == 0
                and (self.in_channels // self.deform_groups) % 8 == 0 and self.adastn.D == self.deform_groups
                and ((self.in_channels // self.deform_groups) // 8) & ((self.in_channels // self.deform_groups) // 8 - 1) == 0
                and nbr.shape == feat_prop.shape)

    def forward(self, nbr_feat_l, ref_feat_l, feat_prop, offset, flag=False):
        n, _, h, w = offset.shape
        if not flag:
            h4, w4 = int(math.floor(h * 0.25)), int(math.floor(w * 0.25))
            h2, w2 = int(math.floor(h * 0.5)), int(math.floor(w * 0.5))
            off_d4 = AG.resize_bilinear_ac(offset, (h4, w4), 0.25)            # :600
            off_d2 = AG.resize_bilinear_ac(offset, (h2, w2), 0.5)             # :601
            # level 3 (:604-608)
            warp4 = AG.flow_warp(nbr_feat_l[2], off_d4)
            p1 = self._level(self.flow_l3, self.trans_l3, warp4, ref_feat_l[2])
            p1_up = AG.resize_bilinear_ac(p1, (2 * h4, 2 * w4), 2.0)
            # level 2 (:609-613)
            warp2 = AG.flow_warp(nbr_feat_l[1], off_d2, flow2=p1_up)
            p2 = self._level(self.flow_l2, self.trans_l2, warp2, ref_feat_l[1])
            p2_up = AG.resize_bilinear_ac(p2, (2 * h2, 2 * w2), 2.0, pre_add=p1_up)
            # level 1 (:614-619)
            warp1 = AG.flow_warp(nbr_feat_l[0], offset, flow2=p2_up)
            p3 = self._level(self.flow_l1, self.trans_l1, warp1, ref_feat_l[0])
            offset = AG.add(p3, p2_up, offset)
        if self._fused_alignment(nbr_feat_l[0], feat_prop, offset):
            # inference hot path: both warps by the refined offset in ONE launch, the second one written in the IL8 layout the
            # DCNv2 kernel samples from; the predictor's 15 D head channels go to that kernel as they are (affine -> offsets
            # and the mask sigmoid, networks.py:302-315, happen in its per-group set-up): de_offset / mask never reach HBM
            if BACKBONE_DTYPE is not None:
                # 16-bit mode (configs[2] / [4]): the warp rounds the sampled features once to bf16 / fp16 (IL8), DCNv2 runs
                # on the 16-bit MFMA with fp32 blend and accumulation; offsets, masks, output stay fp32
                nbr, feat_il = ops.flow_warp_pair(nbr_feat_l[0], feat_prop, offset, b_il8=BACKBONE_DTYPE)
                heads = self.adastn.heads(nbr, ref_feat_l[0])
                return ops.dcnv2_il16(feat_il, heads, None, self.weight, self.bias, self.deform_groups, heads=True)
            nbr, feat_il = ops.flow_warp_pair(nbr_feat_l[0], feat_prop, offset, b_il8=True)          # :621, :623
            act = ops.heads_mask_activated(int(self.weight.shape[1]), self.deform_groups)      # the mask sigmoid in the heads' epilogue (:313-314)
            heads = self.adastn.heads(nbr, ref_feat_l[0], mask_activated=act)                        # :625
            return ops.dcnv2_il(feat_il, heads, None, self.weight, self.bias, self.deform_groups,
                                nprod=int(ops.DCN_MODE[2]), heads=True, mask_activated=act)          # :627-630
        nbr = AG.flow_warp(nbr_feat_l[0], offset)                             # :621
        feat = AG.flow_warp(feat_prop, offset)                                # :623
        de_offset, mask = self.adastn(nbr, ref_feat_l[0])                      # :625
        return modulated_deform_conv2d(feat, de_offset, mask, self.weight, self.bias, self.stride, self.padding,
                                       self.dilation, self.groups, self.deform_groups)   # :627-630


# --------------------------------------------------------------------------------------------
# residual channel-attention backbone (networks.py:432-482)
# --------------------------------------------------------------------------------------------
class CALayer(nn.Module):
    def __init__(self, channel=64, reduction=16):
        super().__init__()
        self.conv_du = nn.Sequential(
            nn.Conv2d(channel, channel // reduction, 1, padding=0, bias=True), _Act("relu"),
            nn.Conv2d(channel // reduction, channel, 1, padding=0, bias=True), _Act("sigmoid"))

    def scale_from_partial(self, partial: Tensor, hw: int) -> Tensor:
        a, b = self.conv_du[0], self.conv_du[2]
        return ops.ca_scale(partial, hw, a.weight, a.bias, b.weight, b.bias)

    def forward(self, x):
        """x * sigmoid(MLP(mean_hw(x))).  Stand-alone form (RCABlock uses the fused path where the
        channel sums come out of the conv epilogue)."""
        n, c, h, w = x.shape
        a, b = self.conv_du[0], self.conv_du[2]
        if AG.needs_grad(x, a.weight, b.weight):
            return AG.rcab_tail(x, torch.zeros_like(x), a.weight, a.bias, b.weight, b.bias)
        partial = x.sum(dim=(2, 3)).view(n, 1, c)
        return ops.scale_residual(x, self.scale_from_partial(partial, h * w), torch.zeros_like(x))


class RCABlock(nn.Module):
    def __init__(self, in_channels=64, out_channels=64, kernel_size=3, stride=1, padding=1, bias=True,
                 mode="CRC", reduction=16):
        super().__init__()
        assert in_channels == out_channels
        if mode[0] in "RL":
            mode = mode[0].lower() + mode[1:]
        self.res = conv(in_channels, out_channels, kernel_size, stride, padding, bias=bias, mode=mode)
        self.ca = CALayer(out_channels, reduction)

    def forward(self, x):
        a, b = self.ca.conv_du[0], self.ca.conv_du[2]
        if AG.needs_grad(x, list(self.parameters())):
            # training: the whole block as one autograd node where it applies (autograd.rcab), else per-op nodes: un-fused mean /
            # MLP / scale so that every piece has its backward kernel
            if (RCAB_ONE_NODE and len(self.res) == 3 and isinstance(self.res[1], _Act) and self.res[1].kind == "relu"
                    and torch.is_grad_enabled()):
                c1, c2 = self.res[0], self.res[2]
                ps = [c1.weight, c1.bias, c2.weight, c2.bias, a.weight, a.bias, b.weight, b.bias]
                if AG.rcab_supported(x, ps):
                    return AG.rcab(x, *ps)
            r = _run_fused(self.res, x)
            return AG.rcab_tail(r, x, a.weight, a.bias, b.weight, b.bias)
        if (RCAB_PRE and x.shape[1] == 64 and len(self.res) == 3 and isinstance(self.res[1], _Act) and self.res[1].kind == "relu"
                and tuple(self.res[2].weight.shape) == (64, 64, 3, 3) and int(a.weight.shape[0]) <= 8):
            # the attention BEFORE the second convolution (ops.ca_scale_pre: its output's channel means are linear in border-corrected
            # channel sums of its input), `res * y + x` as that convolution's epilogue: no scale_residual launch (88 MB of HBM traffic
            # per block at 2 x 64 x 180 x 320)
            c1, c2 = self.res[0], self.res[2]
            # (this branch is inference-only: AG.needs_grad was checked above; the border pieces are a by-product of the first
            # convolution's epilogue where the grouped F(4x4,3x3) kernel runs, None elsewhere: then ca_scale_pre sums the lines itself)
            t, tpart, pieces = ops.conv2d(x, c1.weight, c1.bias, act="relu", chan_partial=True, border=True)
            if not RCAB_PRE_PIECES:
                pieces = None
            scale = ops.ca_scale_pre(t, tpart, c2.weight, c2.bias, a.weight, a.bias, b.weight, b.bias, border=pieces)
            return ops.conv2d(t, c2.weight, c2.bias, residual=x, res_scale=scale)
        r, partial = _run_fused(self.res, x, chan_partial=True)     # conv-ReLU-conv, + channel sums
        if FUSE_CA_TAIL:
            return ops.ca_tail(r, partial, a.weight, a.bias, b.weight, b.bias, x)      # CALayer + res * y + x in one launch
        scale = self.ca.scale_from_partial(partial, x.shape[2] * x.shape[3])
        return ops.scale_residual(r, scale, x)                      # res * y + x  (:463-464)


# Fold `res * y + x` of block k into the first conv of block k+1 (the ca_* fields of the conv descriptor; applied in the
# input transform of the Winograd kernels, in the patch prologue of the direct kernel).  Measured at 4x64x180x320 on the
# F(4x4,3x3) kernel: the fused conv costs +21 us (two input patches, the scale FMA on the transform's critical path, the
# side output) against the 27 us scale_residual launch it removes - 314 ms vs 311 ms per step - so it is off by default.
import os as _os
# The RCAB tail (mean -> MLP -> sigmoid -> res * y + x, networks.py:444-447,463-464) as ONE launch (eavsr_ca_tail_f32) instead of
# ca_scale + scale_residual: EAVSR_FUSE_CA_TAIL=1.  OFF by default.  Round 3's kernel is shaped to run beside a resident
# convolution workgroup (256 threads, 27 registers, 4.6 KB of LDS), is bit-identical and 7.6 us shorter back to back (22.3 against
# 29.9 us per block) -- and the two-stream step is still SLOWER with it, 241.6 -> 249.8 ms (A/B on one box): three 16-register
# scale_residual waves fit per SIMD beside a convolution but one 32-register wave of the fused kernel, each holding its slot through a
# latency-bound MLP prologue, while ca_scale's two workgroups leave the GPU to the other stream.
FUSE_CA_TAIL = _os.environ.get("EAVSR_FUSE_CA_TAIL", "0") == "1"
# fp32 inference: the RCAB's attention before its second convolution, the tail as that convolution's epilogue (as RCAB_H16_PRE in the
# 16-bit modes): 217.0 / 218.0 -> 213.0 / 214.0 ms per configs[1] step in rotation on one box (tools/visits/r5_o.sh), timed output
# bit-identical to the eager forward.  EAVSR_RCAB_PRE=0 keeps conv, conv, ca_scale, scale_residual (A/B switch).
RCAB_PRE = _os.environ.get("EAVSR_RCAB_PRE", "1") == "1"
# round 6: the border-line sums that ca_scale_pre needs come out of the first convolution's epilogue (desc.border_pieces) instead of a
# launch of their own: one small launch per RCAB on the dependent chain instead of two.  EAVSR_RCAB_PRE_PIECES=0: A/B switch.
RCAB_PRE_PIECES = _os.environ.get("EAVSR_RCAB_PRE_PIECES", "1") == "1"


def set_rcab_pre(on: bool) -> None:
    global RCAB_PRE
    RCAB_PRE = bool(on)
# training: one autograd node per RCAB (autograd._RcabFn); EAVSR_RCAB_ONE_NODE=0: the per-op nodes of rounds 1-4
RCAB_ONE_NODE = _os.environ.get("EAVSR_RCAB_ONE_NODE", "1") != "0"
FUSE_CA_INTO_CONV = _os.environ.get("EAVSR_FUSE_CA", "0") == "1"
# One kernel per pyramid level of the residual-flow refinement (eavsr_flow_level_f32: front end + 64 -> 6 heads + affine +
# 18 -> 2 conv, only the two inputs and the 2-channel flow touch HBM) instead of four launches.  Correct (goldens G2 / G5,
# tests/test_hip_ops.py) but OFF by default: measured 28.5 ms per 2-clip forward against 13.0 ms for the four launches --
# a tile's 64 channel steps are serial inside one 4-wave workgroup (4 barriers each) and the 240 tiles of a 2 x 180 x 320
# launch leave no second workgroup per CU to hide them, while the un-fused kernels run 64x more workgroups.  EAVSR_FUSE_LEVEL=1
# / set_fuse_flow_level(True) turns it on (DESIGN.md has the analysis and what a channel-parallel version would need).
FUSE_FLOW_LEVEL = _os.environ.get("EAVSR_FUSE_LEVEL", "0") == "1"


def set_fuse_flow_level(on: bool) -> None:
    global FUSE_FLOW_LEVEL
    if on:
        ops.require_lab("the one-kernel-per-pyramid-level form (eavsr_flow_level_f32)")
    FUSE_FLOW_LEVEL = bool(on)


# In the 16-bit modes the predictor's 5x5 heads run on the 16-bit matrix pipe as well (csrc/conv5_h16.hip); EAVSR_HEADS_16BIT=0
# keeps them on the fp32 Winograd kernel.
HEADS_IN_16BIT = _os.environ.get("EAVSR_HEADS_16BIT", "1") == "1"
# Optional 16-bit residual backbone: None (exact fp32, the default and the BASELINE headline), "bf16" or "fp16"
# (set_backbone_dtype / EAVSR_BACKBONE_DTYPE).  Only the RCAGroup internals change precision.
BACKBONE_DTYPE = _os.environ.get("EAVSR_BACKBONE_DTYPE") or None
# conv -> ReLU -> conv of an RCAB as ONE launch in the 16-bit modes (eavsr_rcab_convs_h16, csrc/rcab_h16.hip: streamed weights, the
# intermediate in LDS; r bit-identical to the two launches).  OPT-IN (EAVSR_RCAB_H16_FUSED=1): back to back it is 49.7 us against
# 52.3 for the two launches at 4 x 64 x 256 x 256 (25.3 against 31.0 at 2 x 180 x 320, 94.1 against 81.7 at 1 x 540 x 960), but
# configs[2]'s two-stream step is 197.3 ms with it against 195.3 without (DESIGN.md 3.6 has the per-phase stamps).
RCAB_H16_FUSED = _os.environ.get("EAVSR_RCAB_H16_FUSED", "0") == "1"


def set_rcab_h16_fused(on: bool) -> None:
    global RCAB_H16_FUSED
    if on:
        ops.require_lab("the one-launch 16-bit RCAB (eavsr_rcab_convs_h16)")
    RCAB_H16_FUSED = bool(on)


# The RCAB's attention computed BEFORE its second convolution and the tail `res * y + x` folded into that convolution's epilogue
# (16-bit modes; csrc/ca.hip eavsr_ca_scale_pre_h16 + eavsr_conv3x3_c64_h16_res): no scale_residual_h16 launch -- three 128-byte-
# per-pixel streams per RCAB.  EAVSR_RCAB_H16_PRE=0 keeps conv, conv, ca_scale, scale_residual (A/B switch).
RCAB_H16_PRE = _os.environ.get("EAVSR_RCAB_H16_PRE", "1") == "1"


def set_rcab_h16_pre(on: bool) -> None:
    global RCAB_H16_PRE
    RCAB_H16_PRE = bool(on)
if BACKBONE_DTYPE is not None:
    ops.set_conv3_h16(BACKBONE_DTYPE)


def set_backbone_dtype(dtype):
    """None -> fp32 everywhere; 'bf16' / 'fp16' -> 16-bit NHWC activations inside every RCAGroup."""
    global BACKBONE_DTYPE
    if dtype is not None:
        ops.h16_code(dtype)
    BACKBONE_DTYPE = dtype
    ops.set_conv3_h16(dtype)      # ... and the plain 3x3 convolutions outside them (first conv of a backbone, encoder): csrc/conv3_h16.hip


import contextlib as _contextlib


@_contextlib.contextmanager
def backbone_dtype(dtype):
    """`with networks.backbone_dtype("bf16"): y = net(x)` -- the 16-bit modes for the body only; the previous setting is
    restored on exit."""
    prev = BACKBONE_DTYPE
    set_backbone_dtype(dtype)
    try:
        yield
    finally:
        set_backbone_dtype(prev)


class RCAGroup(nn.Module):
    def __init__(self, in_channels=64, out_channels=64, kernel_size=3, stride=1, padding=1, bias=True,
                 mode="CRC", reduction=16, nb=12):
        super().__init__()
        assert in_channels == out_channels
        if mode[0] in "RL":
            mode = mode[0].lower() + mode[1:]
        RG = [RCABlock(in_channels, out_channels, kernel_size, stride, padding, bias, mode, reduction)
              for _ in range(nb)]
        RG.append(conv(out_channels, out_channels, mode="C"))
        self.rg = nn.Sequential(*RG)

    def _forward_h16(self, x, dtype):
        """16-bit backbone (BASELINE configs[2] / [4]): NHWC bf16 / fp16 activations through the whole group,
        fp32 accumulation and fp32 channel-attention statistics; fp32 NCHW at the group's boundary."""
        blocks, last = list(self.rg)[:-1], self.rg[-1]
        hw = x.shape[2] * x.shape[3]
        xs = ops.to_nhwc_h16(x, dtype)
        for blk in blocks:
            c1, c2 = blk.res[0], blk.res[2]
            if RCAB_H16_FUSED and ops.rcab_convs_h16_preferred(xs):      # conv -> ReLU -> conv as ONE launch (csrc/rcab_h16.hip)
                r, partial = ops.rcab_convs_h16(xs, c1.weight, c1.bias, c2.weight, c2.bias, chan_partial=True)
            elif RCAB_H16_PRE and int(blk.ca.conv_du[0].weight.shape[0]) <= 8:      # (the kernel's hidden-unit bound)
                # the attention BEFORE the second convolution (its output's channel means are linear in sums of its input), the
                # tail `res * y + x` as that convolution's epilogue: no scale_residual launch (csrc/ca.hip, ops.ca_scale_pre_h16)
                # (round 6: the border-line sums the attention needs come out of the first convolution's epilogue: one small launch
                # per block on the dependent chain instead of two; EAVSR_RCAB_PRE_PIECES=0 keeps the border-sum launch)
                t, tpart, pieces = ops.conv3x3_c64_h16(xs, c1.weight, c1.bias, relu=True, chan_partial=True, border=True) \
                    if RCAB_PRE_PIECES else (*ops.conv3x3_c64_h16(xs, c1.weight, c1.bias, relu=True, chan_partial=True), None)
                du_a, du_b = blk.ca.conv_du[0], blk.ca.conv_du[2]
                scale = ops.ca_scale_pre_h16(t, tpart, c2.weight, c2.bias, du_a.weight, du_a.bias, du_b.weight, du_b.bias, border=pieces)
                xs = ops.conv3x3_c64_h16_res(t, c2.weight, c2.bias, xs, scale)
                continue
            else:
                t = ops.conv3x3_c64_h16(xs, c1.weight, c1.bias, relu=True)
                r, partial = ops.conv3x3_c64_h16(t, c2.weight, c2.bias, chan_partial=True)
            xs = ops.scale_residual_h16(r, blk.ca.scale_from_partial(partial, hw), xs)
        y = ops.conv3x3_c64_h16(xs, last.weight, last.bias)
        return ops.from_nhwc_h16(y, residual=x)

    def forward(self, x):
        blocks, last = list(self.rg)[:-1], self.rg[-1]
        if BACKBONE_DTYPE is not None and not AG.needs_grad(x, list(self.parameters())) and x.shape[1] == 64 and \
                all(isinstance(b, RCABlock) and len(b.res) == 3 and isinstance(b.res[1], _Act) and b.res[1].kind == "relu"
                    and tuple(b.res[0].weight.shape) == (64, 64, 3, 3) for b in blocks) and \
                tuple(last.weight.shape) == (64, 64, 3, 3):
            return self._forward_h16(x, BACKBONE_DTYPE)
        fuse = FUSE_CA_INTO_CONV and not AG.needs_grad(x, list(self.parameters())) and ops.ca_fusable(x, last.weight.shape[0]) and len(blocks) > 0 and all(
            isinstance(b, RCABlock) and len(b.res) == 3 and isinstance(b.res[1], _Act) and b.res[1].kind == "relu"
            for b in blocks)
        if not fuse:
            r = x
            for blk in blocks:
                r = blk(r)
            return last(r, residual=x)                              # conv, + x  (:480-482)
        # Fused chain: the tail `res * y + x` of block k (networks.py:463-464) is applied inside the first
        # conv of block k+1 (and inside the group's last conv), which also emits the new residual stream.
        hw = x.shape[2] * x.shape[3]
        xs, r, scale = x, None, None
        for blk in blocks:
            c1, c2 = blk.res[0], blk.res[2]
            if r is None:
                t = c1(xs, act="relu")
            else:
                t, xs = ops.conv2d(r, c1.weight, c1.bias, act="relu", ca=(scale, xs), ca_out=True)
            r, partial = c2(t, chan_partial=True)
            scale = blk.ca.scale_from_partial(partial, hw)
        return ops.conv2d(r, last.weight, last.bias, residual=x, ca=(scale, xs))


# --------------------------------------------------------------------------------------------
# encoder (networks.py:522-552) -- a caller of the path ("next" row f2), run on the same conv kernel
# --------------------------------------------------------------------------------------------
class ContrasExtractorLayer(nn.Module):
    """VGG16 conv1_1 .. conv3_1 without the two pools + tail conv.  The reference pulls
    torchvision's pretrained VGG16 at construction; torchvision is not part of this image, so the
    convs start from PyTorch's default init and real weights arrive through load_state_dict
    (same keys: encoder.model.conv{1_1,1_2,2_1,2_2,3_1}.*, encoder.tail.*)."""

    def __init__(self, n_feat=64):
        super().__init__()
        cfg = [("conv1_1", 3, 64), ("conv1_2", 64, 64), ("conv2_1", 64, 128), ("conv2_2", 128, 128),
               ("conv3_1", 128, 256)]
        od = OrderedDict()
        for i, (name, ci, co) in enumerate(cfg):
            od[name] = Conv2d(ci, co, 3, 1, 1)
            if i < len(cfg) - 1:
                od[name.replace("conv", "relu")] = _Act("relu")
        self.tail = Conv2d(256, n_feat, 3, 1, 1)
        self.model = nn.Sequential(od)
        self.register_buffer("mean", torch.Tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer("std", torch.Tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

    def forward(self, batch):
        batch = (batch - self.mean) / self.std if batch.requires_grad else ops.normalize(batch, self.mean, self.std)
        return self.tail(_run_fused(self.model, batch))


def init_net(net, init_type="default", init_gain=0.02, gpu_ids=()):
    """networks.py:67-74 replacement: one process per GPU, so no DataParallel wrapper -- the module
    moves to gpu_ids[0] (or the current device) and is returned as is."""
    if init_type not in ("default", None):
        raise NotImplementedError("only init_type='default' (what the EAVSR scripts use) is supported")
    if len(gpu_ids) > 0:
        net.to(torch.device("cuda", gpu_ids[0]))
    return net
