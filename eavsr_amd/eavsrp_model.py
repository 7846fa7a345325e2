"""Drop-in counterparts of models/eavsrp_model.py (x4) and models/eavsrpx2_model.py (x2).

`EAVSRP(opt, spynet_pretrained=None).forward(lrs)` keeps the reference signature
((n,t,3,h,w) in [0,1] -> (n,t,3,s*h,s*w), eavsrp_model.py:202-240) and state_dict keys, and runs
the whole forward on libeavsr_hip.so kernels: the alignment / propagation hot path (MultiAdSTN,
DCNv2, flow_warp, fusion, 30-RCAB backbones) and, on the same conv kernel, the callers either
side of it (SPyNet, encoder, upsampling tail).  torch is used for memory, views, and a few
elementwise / pooling glue ops outside the hot loop.

One process drives one GPU (no nn.DataParallel: networks.init_net here does not wrap).
"""
from __future__ import annotations

import os
import time
from typing import Dict, List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import autograd as AG
from . import networks as N
from . import ops

# EAVSR_PREPACK=0: every weight's packed forms at their first use (the A/B switch of ops.prepack_conv3_x6)
PREPACK = os.environ.get("EAVSR_PREPACK", "1") != "0"

# In the 16-bit modes the upsampling tail (conv + PixelShuffle stages, conv_hr, conv_last) runs on the 16-bit kernels as well;
# EAVSR_TAIL_16BIT=0 keeps it on the fp32 kernels (A/B switch).
TAIL_IN_16BIT = os.environ.get("EAVSR_TAIL_16BIT", "1") == "1"

Tensor = torch.Tensor


# ---------------------------------------------------------------------------------------------
# flow_warp with the flow in NHWC (eavsrp_model.py:587-626)
# ---------------------------------------------------------------------------------------------
def flow_warp(x, flow, interpolation="bilinear", padding_mode="zeros", align_corners=True):
    """Warp `x` (n,c,h,w) by `flow` (n,h,w,2): [...,0] = x displacement, [...,1] = y, in pixels."""
    return AG.flow_warp(x, flow, padding_mode=padding_mode, flow_layout="nhwc", interpolation=interpolation,
                        align_corners=align_corners)


# ---------------------------------------------------------------------------------------------
# residual backbone (eavsrp_model.py:366-400)
class _FrameList(list):
    """a list of per-frame tensors (n, c, h, w) that may know the frame-major tensor (t*n, c, h, w) its items are views of"""

    def __init__(self, items, stacked=None):
        super().__init__(items)
        self.stacked = stacked


def _frame_major(frames) -> Tensor:
    st = getattr(frames, "stacked", None)
    return st if st is not None else torch.cat(list(frames), 0)


# ---------------------------------------------------------------------------------------------
class ResidualBlocksWithInputConv(nn.Module):
    """conv3x3(in -> out) + LeakyReLU(0.1) -> RCAGroup(nb=num_blocks).  `feat` may be a tensor or a
    list of tensors standing for their channel concatenation (the cat at eavsrp_model.py:322,353 is
    never materialised)."""

    def __init__(self, in_channels, out_channels=64, num_blocks=30):
        super().__init__()
        self.main = nn.Sequential(
            N.Conv2d(in_channels, out_channels, 3, 1, 1, bias=True),
            N._Act("lrelu", 0.1),
            N.RCAGroup(out_channels, out_channels, nb=num_blocks))

    def forward(self, feat):
        x = self.main[0](feat, act="lrelu", slope=0.1)
        return self.main[2](x)


# ---------------------------------------------------------------------------------------------
# SPyNet (eavsrp_model.py:402-585) -- caller of the path ("next" row f1); uses flow_warp(border)
# ---------------------------------------------------------------------------------------------
class ConvModule(nn.Module):
    """Key-compatible with mmcv.cnn.ConvModule(norm_cfg=None): .conv (+ fused ReLU)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, norm_cfg=None,
                 act_cfg=dict(type="ReLU")):
        super().__init__()
        assert norm_cfg is None and stride == 1 and padding == kernel_size // 2
        self.conv = N.Conv2d(in_channels, out_channels, kernel_size, stride, padding)
        self.with_activation = act_cfg is not None
        if self.with_activation:
            assert act_cfg["type"] == "ReLU"
            self.activate = N._Act("relu")

    def forward(self, x):
        return self.conv(x, act="relu" if self.with_activation else None)


class SPyNetBasicModule(nn.Module):
    def __init__(self):
        super().__init__()
        chans = [(8, 32), (32, 64), (64, 32), (32, 16), (16, 2)]
        self.basic_module = nn.Sequential(*[
            ConvModule(ci, co, 7, 1, 3, norm_cfg=None, act_cfg=dict(type="ReLU") if i < 4 else None)
            for i, (ci, co) in enumerate(chans)])

    def forward(self, tensor_input):
        return self.basic_module(tensor_input)


def load_checkpoint(module: nn.Module, path: str, strict: bool = True):
    """Reader for mmcv-style checkpoints ({'state_dict': ...} or a bare state dict), as
    mmcv.runner.load_checkpoint is used at eavsrp_model.py:421."""
    ck = torch.load(path, map_location="cpu")
    sd = ck.get("state_dict", ck) if isinstance(ck, dict) else ck
    sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
    module.load_state_dict(sd, strict=strict)
    return ck


class SPyNet(nn.Module):
    def __init__(self, pretrained):
        super().__init__()
        self.basic_module = nn.ModuleList([SPyNetBasicModule() for _ in range(6)])
        if isinstance(pretrained, str):
            load_checkpoint(self, pretrained, strict=True)
        elif pretrained is not None:
            raise TypeError(f"[pretrained] should be str or None, but got {type(pretrained)}.")
        self.register_buffer("mean", torch.Tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer("std", torch.Tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

    def compute_flow(self, ref, supp):
        """eavsrp_model.py:433-488"""
        n, _, h, w = ref.size()
        ref = [ops.normalize(ref, self.mean, self.std)]
        supp = [ops.normalize(supp, self.mean, self.std)]
        for _ in range(5):
            ref.append(ops.avg_pool2(ref[-1]))
            supp.append(ops.avg_pool2(supp[-1]))
        ref, supp = ref[::-1], supp[::-1]
        flow = ref[0].new_zeros(n, 2, h // 32, w // 32)
        for level in range(len(ref)):
            if level == 0:
                flow_up = flow
            else:
                flow_up = ops.resize_bilinear_ac(flow, (flow.shape[2] * 2, flow.shape[3] * 2), 2.0)
            warped = ops.flow_warp(supp[level], flow_up, padding_mode="border")
            res = self.basic_module[level](ops.concat3(ref[level], warped, flow_up))
            flow = ops.add(flow_up, res)
        return flow

    def forward(self, ref, supp):
        """eavsrp_model.py:490-523"""
        h, w = ref.shape[2:4]
        w_up = w if (w % 32) == 0 else 32 * (w // 32 + 1)
        h_up = h if (h % 32) == 0 else 32 * (h // 32 + 1)
        # F.interpolate(.., size, 'bilinear', align_corners=False) both ways and the flow rescaling, as HIP kernels; a resize to
        # the same size is the identity (source coordinate = destination coordinate, weights 1 / 0) and is skipped
        if (h_up, w_up) != (h, w):
            ref = ops.resize_bilinear(ref, (h_up, w_up))
            supp = ops.resize_bilinear(supp, (h_up, w_up))
        flow = self.compute_flow(ref, supp)
        if (h_up, w_up) != (h, w):
            flow = ops.resize_bilinear(flow, (h, w), channel_mul=(float(w) / float(w_up), float(h) / float(h_up)))
        return flow


# ---------------------------------------------------------------------------------------------
# EAVSRP
# ---------------------------------------------------------------------------------------------
_PYR = ("spatial", "spatial_d2", "spatial_d4")


class EAVSRP(nn.Module):
    """eavsrp_model.py:121-364 (scale 4) / eavsrpx2_model.py:124-365 (scale 2, selected by
    opt.scale or the `scale` argument)."""

    def __init__(self, opt, spynet_pretrained=None, scale=None):
        super().__init__()
        self.opt = opt
        self.predict = getattr(opt, "predict", False)
        self.n_resblock = 30
        self.n_frame = getattr(opt, "n_frame", 7)
        self.n_feats = 64
        self.n_flow = getattr(opt, "n_flow", 5)
        self.scale = int(scale if scale is not None else getattr(opt, "scale", 4))
        if self.scale not in (2, 4):
            raise NotImplementedError("EAVSRP exists for x4 (eavsrp_model) and x2 (eavsrpx2_model)")

        self.spynet = SPyNet(pretrained=spynet_pretrained)
        for p in self.spynet.parameters():
            p.requires_grad = False
        self.encoder = N.ContrasExtractorLayer(self.n_feats)

        self.deform_align = nn.ModuleDict()
        self.backbone = nn.ModuleDict()
        self.fusion = nn.ModuleDict()
        for i, module in enumerate(["backward_1", "forward_1", "backward_2", "forward_2"]):
            self.deform_align[module] = N.MultiAdSTN(opt, self.n_feats, self.n_feats, deformable_groups=8)
            self.backbone[module] = ResidualBlocksWithInputConv((2 + i) * self.n_feats, self.n_feats, self.n_resblock)
            self.fusion[module] = N.Conv2d(self.n_feats * 3, self.n_feats, 1, 1, 0, bias=True)

        self.reconstruction = ResidualBlocksWithInputConv(5 * self.n_feats, self.n_feats, 5)
        self.upsample1 = N.seq([N.conv(self.n_feats, self.n_feats * 4, mode="C"), nn.PixelShuffle(2)])
        if self.scale == 4:
            self.upsample2 = N.seq([N.conv(self.n_feats, self.n_feats * 4, mode="C"), nn.PixelShuffle(2)])
        self.conv_hr = N.Conv2d(64, 64, 3, 1, 1)
        self.conv_last = N.Conv2d(64, 3, 3, 1, 1)
        self.img_upsample = nn.Upsample(scale_factor=self.scale, mode="bilinear", align_corners=False)
        self.lrelu = N._Act("lrelu", 0.1)

    # -- flows -----------------------------------------------------------------------------
    def compute_flow(self, lrs, lr_tm=None):
        """eavsrp_model.py:179-200.  Both directions go through SPyNet as one batch."""
        n, t, c, h, w = lrs.shape
        # frame-major pairs (pair (frame i, clip b) -> row i n + b): the two frame runs are contiguous views of the frame-major
        # input, and a time step's flow `flows[:, i]` of the (n, t-1, 2, h, w) VIEWS returned here is contiguous, so that
        # `propagate` copies nothing (44 strided copies per forward in round 2)
        lr_tm = lrs.transpose(0, 1).reshape(t * n, c, h, w) if lr_tm is None else lr_tm
        m = n * (t - 1)
        lrs_1, lrs_2 = lr_tm[:m], lr_tm[n:]
        both = self.spynet(torch.cat([lrs_1, lrs_2], 0), torch.cat([lrs_2, lrs_1], 0))
        flows_backward = both[:m].view(t - 1, n, 2, h, w).transpose(0, 1)
        flows_forward = both[m:].view(t - 1, n, 2, h, w).transpose(0, 1)
        return flows_forward, flows_backward

    # -- forward ---------------------------------------------------------------------------
    def forward(self, lrs):
        n, t, c, h, w = lrs.shape
        assert h >= 64 and w >= 64, (
            'The height and width of inputs should be at least 64, '
            f'but got {h} and {w}.')
        if not lrs.is_cuda:
            raise RuntimeError("eavsr_amd.EAVSRP runs on the GPU only (no CPU path); for a CPU reference use "
                               "the original repository with --gpu_ids -1")
        if torch.is_grad_enabled() and PREPACK and ops.x6s_takes(n, h, w):
            # training at a crop: the packed forms (forward + input-gradient) of every trainable 3x3 64 -> 64 weight in a handful of
            # launches here instead of one launch per weight and form at its first use (ops.prepack_conv3_x6)
            ops.prepack_conv3_x6([p_ for p_ in self.parameters() if p_.requires_grad and p_.dim() == 4 and tuple(p_.shape) == (64, 64, 3, 3)])
        # frame-major input so that every per-frame slice (of the input, the features, the flows) is contiguous
        lr_tm = lrs.transpose(0, 1).reshape(t * n, c, h, w)
        with torch.no_grad():
            flows_forward, flows_backward = self.compute_flow(lrs, lr_tm)

        f1 = self.encoder(lr_tm)                                   # :216
        f2, f4 = AG.pyramid(f1)                                    # :218-220
        feats: Dict[str, List[Tensor]] = {
            "spatial": _FrameList([f1[i * n:(i + 1) * n] for i in range(t)], f1),
            "spatial_d2": [f2[i * n:(i + 1) * n] for i in range(t)],
            "spatial_d4": [f4[i * n:(i + 1) * n] for i in range(t)],
        }
        for iter_ in (1, 2):
            for direction in ("backward", "forward"):
                module = f"{direction}_{iter_}"
                feats[module] = []
                flows = flows_backward if direction == "backward" else flows_forward
                feats = self.propagate(feats, flows, module)
        return self.upsample(lrs, feats, lr_tm)

    def propagate(self, feats, flows, module_name):
        """eavsrp_model.py:242-329."""
        n, t, _, h, w = flows.size()
        frame_idx = list(range(0, t + 1))
        flow_idx = list(range(-1, t))
        mapping_idx = list(range(0, len(feats["spatial"])))
        mapping_idx += mapping_idx[::-1]
        backward = "backward" in module_name
        if backward:
            frame_idx = frame_idx[::-1]
            flow_idx = frame_idx
        step = 1 if backward else -1
        align, fusion, backbone = self.deform_align[module_name], self.fusion[module_name], self.backbone[module_name]
        feat_prop = flows.new_zeros(n, self.n_feats, h, w)
        zeros = None
        # inference: the branch's features go straight into one frame-major buffer (frame idx -> rows idx*n ..), which is what the
        # tail consumes -- the list below holds views of it and `upsample` needs no torch.cat (5 x 0.4 GB per 2-clip forward)
        stacked = None
        if not AG.needs_grad(flows, feats["spatial"][0], list(backbone.parameters())):
            stacked = flows.new_empty(len(feats["spatial"]) * n, self.n_feats, h, w)
        for i, idx in enumerate(frame_idx):
            cur = [feats[k][mapping_idx[idx]] for k in _PYR]
            if i > 0:
                nbr = [feats[k][mapping_idx[idx + step]] for k in _PYR]
                flow_n1 = flows[:, flow_idx[i]].contiguous()
                cond_n1 = align(nbr, cur, feat_prop, flow_n1)
                if i > 1:
                    feat_n2 = feats[module_name][-2]
                    nbr2 = [feats[k][mapping_idx[idx + 2 * step]] for k in _PYR]
                    flow_n2 = flows[:, flow_idx[i - 1]].contiguous()
                    flow_n2 = AG.add(flow_n1, AG.flow_warp(flow_n2, flow_n1))            # :309-310
                    cond_n2 = align(nbr2, cur, feat_n2, flow_n2)
                else:
                    if zeros is None:
                        zeros = torch.zeros_like(cond_n1)
                    cond_n2 = zeros
                feat_prop = fusion([cond_n1, cur[0], cond_n2])                          # :313-314
            others = [feats[k][idx] for k in feats if k not in _PYR and k != module_name]
            res = backbone([cur[0]] + others + [feat_prop])                             # :317-323
            # AG.add honours `out` only without autograd: as soon as one frame's sum carries a gradient (a frozen backbone
            # behind a trained alignment / fusion module, an earlier branch with gradients) the buffer is dropped for the whole
            # branch -- its rows would stay uninitialised and cut off from autograd -- and `upsample` concatenates the list
            if stacked is not None and AG.needs_grad(feat_prop, res):
                stacked = None
            feat_prop = AG.add(feat_prop, res, out=None if stacked is None else stacked[idx * n:(idx + 1) * n])
            feats[module_name].append(feat_prop)
        if backward:
            feats[module_name] = feats[module_name][::-1]
        feats[module_name] = _FrameList(feats[module_name], stacked)
        return feats

    def upsample(self, lqs, feats, lq_tm=None):
        """eavsrp_model.py:331-364, all t frames as one batch."""
        n, t = lqs.shape[:2]
        branches = [k for k in feats if k not in _PYR]
        srcs = [_frame_major(feats["spatial"])] + [_frame_major(feats[k]) for k in branches]
        hr = self.reconstruction(srcs)
        if lq_tm is None:
            lq_tm = lqs.transpose(0, 1).reshape(t * n, *lqs.shape[2:])
        tail16 = (N.BACKBONE_DTYPE is not None and TAIL_IN_16BIT and self.n_feats == 64 and
                  not AG.needs_grad(hr, lq_tm, list(self.upsample1.parameters()) + list(self.conv_hr.parameters()) +
                                    list(self.conv_last.parameters()) + (list(self.upsample2.parameters()) if self.scale == 4 else [])))
        if tail16:
            # 16-bit modes (BASELINE configs[2] / [4]): the tail on the 16-bit backbone kernel -- every conv + PixelShuffle(2) stage as
            # four 64 -> 64 slices whose store pattern is the shuffle, conv_hr as it is, conv_last by packed dot products; the
            # activations between them 16-bit NHWC, the skip image and the result fp32
            hr = ops.to_nhwc_h16(hr, N.BACKBONE_DTYPE)
            u1 = self.upsample1[0]
            hr = ops.conv3x3_c64_h16_act(hr, u1.weight, u1.bias, act="lrelu", slope=0.1, pixel_shuffle2=True)
            if self.scale == 4:
                u2 = self.upsample2[0]
                hr = ops.conv3x3_c64_h16_act(hr, u2.weight, u2.bias, act="lrelu", slope=0.1, pixel_shuffle2=True)
            hr = ops.conv3x3_c64_h16_act(hr, self.conv_hr.weight, self.conv_hr.bias, act="lrelu", slope=0.1)
            skip = ops.resize_bilinear(lq_tm, (self.scale * lq_tm.shape[2], self.scale * lq_tm.shape[3]))
            out = ops.conv3x3_c64to3_h16(hr, self.conv_last.weight, self.conv_last.bias, residual=skip)
            return out.view(t, n, *out.shape[1:]).transpose(0, 1).contiguous()
        # conv -> PixelShuffle(2) -> LeakyReLU (:343-347): the activation commutes with the shuffle and the shuffle is the conv
        # kernel's own store pattern (the torch copy was 0.8 / 3.3 GB per 2-clip forward)
        hr = self.upsample1[0](hr, act="lrelu", slope=0.1, pixel_shuffle2=True)
        if self.scale == 4:
            hr = self.upsample2[0](hr, act="lrelu", slope=0.1, pixel_shuffle2=True)
        hr = self.conv_hr(hr, act="lrelu", slope=0.1)
        # nn.Upsample(scale_factor, 'bilinear', align_corners=False) (:158,359): source coordinate (dst + 0.5) / s - 0.5
        skip = (self.img_upsample(lq_tm) if lq_tm.requires_grad else
                ops.resize_bilinear(lq_tm, (self.scale * lq_tm.shape[2], self.scale * lq_tm.shape[3])))
        out = self.conv_last(hr, residual=skip)                                      # :359-360
        return out.view(t, n, *out.shape[1:]).transpose(0, 1).contiguous()


class EAVSRPx2(EAVSRP):
    """eavsrpx2_model.py's network (class EAVSRP there): one pixel-shuffle stage, x2 bilinear skip."""

    def __init__(self, opt, spynet_pretrained=None):
        super().__init__(opt, spynet_pretrained, scale=2)


# ---------------------------------------------------------------------------------------------
# model wrapper (eavsrp_model.py:18-119), inference side
# ---------------------------------------------------------------------------------------------
def get_scheduler(optimizer, opt):
    """models/networks.py:16-37: the four learning-rate policies of the reference's options, same constants."""
    from torch.optim import lr_scheduler
    policy = getattr(opt, "lr_policy", "step")
    if policy == "linear":
        def lambda_rule(epoch):
            return 1 - max(0, epoch - opt.niter) / max(1, float(opt.niter_decay))
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda_rule)
    if policy == "step":
        return lr_scheduler.StepLR(optimizer, step_size=opt.lr_decay_iters, gamma=0.5)
    if policy == "plateau":
        return lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", factor=0.2, threshold=0.01, patience=5)
    if policy == "cosine":
        return lr_scheduler.CosineAnnealingLR(optimizer, T_max=opt.niter, eta_min=0)
    raise NotImplementedError("lr [%s] is not implemented" % policy)


def make_optimizer(net: "EAVSRP", opt) -> torch.optim.Adam:
    """Adam with the reference's two parameter groups (eavsrp_model.py:45-59): the alignment modules at lr 1e-5, the
    rest at opt.lr -- both built from ALL of `net.parameters()` in registration order exactly as the reference builds
    them, the 60 frozen SPyNet tensors included.  Those never receive a gradient, so Adam skips them, but they count in
    the state_dict's parameter indices: an `EAVSRP_optimizer_Adam.pth` written by either implementation resumes in the
    other (base_model.py:251-270)."""
    align_ids = {id(p) for p in net.deform_align.parameters()}
    every = list(net.parameters())
    basic = [p for p in every if id(p) not in align_ids]
    align = [p for p in every if id(p) in align_ids]
    return torch.optim.Adam([{"params": basic}, {"params": align, "lr": 1e-5}], lr=getattr(opt, "lr", 1e-4),
                            betas=(getattr(opt, "beta1", 0.9), getattr(opt, "beta2", 0.999)),
                            weight_decay=getattr(opt, "weight_decay", 0.0))


class EAVSRPModel:
    """Stand-in of EAVSRPModel / EAVSRPx2Model (models/eavsrp_model.py:18-119): set_input / forward / test /
    optimize_parameters / get_current_visuals / load_networks / save_networks with the reference's
    checkpoint format ({'state_dict': ...}, base_model.py:159-217).  One process per GPU; with several
    processes (torch.distributed initialised) the training step averages the loss gradients with one
    bucketed RCCL all-reduce (eavsr_amd.shard.GradientAllReducer) instead of nn.DataParallel.
    The reference's late-training PWC-Net mask (epoch >= opt.npost, eavsrp_model.py:85-97) is out of scope."""

    def __init__(self, opt):
        self.opt = opt
        self.scale = opt.scale
        self.isTrain = getattr(opt, "isTrain", False)
        gpu_ids = list(getattr(opt, "gpu_ids", [0]))
        if len(gpu_ids) == 0:
            raise RuntimeError("eavsr_amd has no CPU path (--gpu_ids -1 is the reference's CPU mode)")
        self.device = torch.device("cuda", gpu_ids[0])
        self.visual_names = ["data_lr_seq", "data_hr_seq", "data_sr_seq"]
        self.loss_names = ["EAVSRP_L1", "EAVSRP_Total"]
        self.model_names = ["EAVSRP"]
        self.netEAVSRP = N.init_net(EAVSRP(opt, getattr(opt, "spynet_pretrained", None)), gpu_ids=gpu_ids)
        self.time, self.isfirst, self.num = 0.0, True, 0
        self.epoch = 0
        self.start_epoch = 0
        self.metric = 0            # learning-rate policy 'plateau' (base_model.py:34)
        self.optimizers, self.schedulers = [], []
        if self.isTrain:
            self.optimizer_EAVSRP = make_optimizer(self.netEAVSRP, opt)
            trainable = [p for p in self.netEAVSRP.parameters() if p.requires_grad]   # what the gradient all-reduce carries
            self.optimizers = [self.optimizer_EAVSRP]
            from .shard import GradientAllReducer
            self.grad_sync = GradientAllReducer(trainable)
            self.netEAVSRP.train()
        self.sync_parameters()

    def sync_parameters(self, src: int = 0) -> int:
        """Several processes (one per GPU): every rank takes rank `src`'s parameters and buffers, once -- at construction and
        after `load_networks` -- so that data-parallel replicas start identical whatever each rank seeded or loaded.  The
        reference gets this from nn.DataParallel's per-forward replication (models/networks.py:67-74); here it is one start-up
        broadcast and nothing per step (eavsr_amd.shard.broadcast_module).  No-op on one process."""
        from .shard import broadcast_module
        return broadcast_module(self.netEAVSRP, src)

    def setup(self, opt=None):
        """base_model.py:56-70, what train_basic.py:42 / test_basic.py call after construction: learning-rate schedulers
        (training), then the checkpoint named by opt.load_iter / opt.load_path and, with opt.load_optimizers, the optimizer
        file."""
        opt = opt if opt is not None else self.opt
        load_iter = getattr(opt, "load_iter", 0)
        if self.isTrain:
            self.schedulers = [get_scheduler(optimizer, opt) for optimizer in self.optimizers]
            for scheduler in self.schedulers:
                scheduler.last_epoch = load_iter
        if (isinstance(load_iter, int) and load_iter > 0) or getattr(opt, "load_path", "") != "":
            self.load_networks(load_iter)
            if getattr(opt, "load_optimizers", False):
                self.load_optimizers(load_iter)
        self.print_networks(getattr(opt, "verbose", False))

    def update_learning_rate(self):
        """base_model.py:131-138"""
        for i, scheduler in enumerate(self.schedulers):
            if scheduler.__class__.__name__ == "ReduceLROnPlateau":
                scheduler.step(getattr(self, "metric", 0))
            else:
                scheduler.step()
            print("lr of %s = %.7f" % (self.optimizer_names[i], scheduler.get_last_lr()[0]))

    def print_networks(self, verbose=False):
        """base_model.py:272-284"""
        print("---------- Networks initialized -------------")
        num_params = sum(p.numel() for p in self.netEAVSRP.parameters())
        if verbose:
            print(self.netEAVSRP)
        print("[Network %s] Total number of parameters : %.3f M" % (self.model_names[0], num_params / 1e6))
        print("-----------------------------------------------")

    def eval(self):
        self.netEAVSRP.eval()

    def train(self):
        self.netEAVSRP.train()

    def set_input(self, input, epoch=0):
        self.data_lr_seq = input["lr_seq"].to(self.device)
        self.data_hr_seq = input["hr_seq"].to(self.device) if "hr_seq" in input else None
        self.image_name = input.get("fname")
        self.idx = min(self.opt.n_frame // 2, self.data_lr_seq.shape[1] - 1)
        self.epoch = epoch

    def forward(self):
        if self.isTrain and self.netEAVSRP.training:
            if self.epoch >= getattr(self.opt, "npost", 350):
                raise NotImplementedError("the PWC-Net validity mask of epochs >= npost (eavsrp_model.py:85-97) is "
                                          "outside the hot path")
            self.data_sr_seq = self.netEAVSRP(self.data_lr_seq)
            self.data_sr = self.data_sr_seq[:, self.idx]
            return
        start = time.time()
        self.data_sr_seq = self.netEAVSRP(self.data_lr_seq)
        self.data_sr = self.data_sr_seq[:, self.idx]
        end = time.time()
        if not self.isfirst:
            self.time += end - start
            self.num += 1
        self.isfirst = False

    def test(self):
        with torch.no_grad():
            self.forward()

    def backward(self):
        # criterionL1(hr, sr).mean()  (eavsrp_model.py:109-113)
        self.loss_EAVSRP_L1 = (self.data_hr_seq - self.data_sr_seq).abs().mean()
        self.loss_EAVSRP_Total = self.loss_EAVSRP_L1
        with AG.grad_sink():         # per-use parameter gradients are summed in place by the wgrad kernels
            self.loss_EAVSRP_Total.backward()
        self.grad_sync.finish()      # one bucketed all-reduce over the loss gradients (no-op on 1 process)

    def optimize_parameters(self):
        self.forward()
        self.optimizer_EAVSRP.zero_grad(set_to_none=True)
        self.backward()
        self.optimizer_EAVSRP.step()

    def get_current_losses(self):
        return {n: float(getattr(self, "loss_" + n).detach()) for n in self.loss_names if hasattr(self, "loss_" + n)}

    def get_current_visuals(self):
        out = {}
        for name in self.visual_names:
            v = getattr(self, name, None)
            if v is not None:
                out[name] = torch.clamp(v.detach() * 255.0, 0, 255).round()
        return out

    # ---- checkpoint and optimizer files (base_model.py:159-270) ---------------------------------
    @property
    def optimizer_names(self):
        return ["EAVSRP_optimizer_%s" % getattr(self.opt, "optimizer", "Adam")]   # eavsrp_model.py:31

    def _save_dir(self):
        d = getattr(self, "save_dir", None) or os.path.join(getattr(self.opt, "checkpoints_dir", "./ckpt"),
                                                             getattr(self.opt, "name", "eavsr"))
        os.makedirs(d, exist_ok=True)
        return d

    def _network_path(self, epoch_or_path):
        """An int is the reference's epoch (`<save_dir>/<name>_model_<epoch>.pth`, base_model.py:161-162); opt.load_path
        overrides it on load (base_model.py:182-183); a string is taken as the file itself."""
        if isinstance(epoch_or_path, int):
            return os.path.join(self._save_dir(), "%s_model_%d.pth" % (self.model_names[0], epoch_or_path))
        return epoch_or_path

    def save_networks(self, epoch_or_path):
        """{'state_dict': cpu tensors} (base_model.py:165-170); with an epoch number the optimizer files are written
        beside it as the reference does (base_model.py:172)."""
        path = self._network_path(epoch_or_path)
        torch.save({"state_dict": {k: v.detach().cpu() for k, v in self.netEAVSRP.state_dict().items()}}, path)
        if isinstance(epoch_or_path, int) and self.isTrain:
            self.save_optimizers(epoch_or_path)
        return path

    def load_networks(self, epoch_or_path):
        """Every key of the file must exist in the network with the same shape and every parameter of the network
        must be in the file: the reference `exit()`s on either mismatch (base_model.py:193-213); here it raises.
        Several processes: a COLLECTIVE -- every rank calls it (as every rank runs train_basic.py / test_basic.py).  The ranks
        first agree that all of them read and checked their file (`shard.all_ranks_ok`): if one failed, every rank raises --
        nobody is left blocked in the parameter broadcast that follows, and that broadcast can never pair with another rank's
        next collective.  Then rank 0's parameters and buffers go to everybody (`sync_parameters`)."""
        from .shard import all_ranks_ok
        path, err = None, None
        try:
            path = self._network_path(epoch_or_path)
            if isinstance(epoch_or_path, int) and getattr(self.opt, "load_path", ""):
                path = self.opt.load_path
            sd = torch.load(path, map_location="cpu")
            sd = sd.get("state_dict", sd)
            own = self.netEAVSRP.state_dict()
            unknown = [k for k in sd if k not in own]
            missing = [k for k in own if k not in sd]
            if unknown:
                raise RuntimeError("Saved parameter named [%s] is not in the network (%d such keys)" % (unknown[0], len(unknown)))
            if missing:
                raise RuntimeError("Parameter named [%s] is not in %s (%d such keys)" % (missing[0], path, len(missing)))
            for k, v in sd.items():
                if tuple(own[k].shape) != tuple(v.shape):
                    raise RuntimeError("While copying the parameter named [%s], whose dimensions in the model are %s and "
                                       "whose dimensions in the checkpoint are %s." % (k, list(own[k].shape), list(v.shape)))
            self.netEAVSRP.load_state_dict(sd, strict=True)
        except Exception as e:      # noqa: BLE001 -- re-raised below, on every rank
            err = e
        if not all_ranks_ok(err is None):
            if err is not None:
                raise err
            raise RuntimeError(f"load_networks({epoch_or_path!r}): another rank failed to load its checkpoint; aborting on every rank")
        self.sync_parameters()
        if isinstance(epoch_or_path, int):
            self.start_epoch = epoch_or_path
        return path

    def save_optimizers(self, epoch):
        """One file per optimizer: {'name', 'epoch', 'state_dict'} at <save_dir>/<optimizer name>.pth (base_model.py:251-260)."""
        assert len(self.optimizers) == len(self.optimizer_names)
        for name, optimizer in zip(self.optimizer_names, self.optimizers):
            torch.save({"name": name, "epoch": epoch, "state_dict": optimizer.state_dict()},
                       os.path.join(self._save_dir(), name + ".pth"))

    def load_optimizers(self, epoch):
        """base_model.py:262-270: name and epoch recorded in the file must match."""
        assert len(self.optimizers) == len(self.optimizer_names)
        for name, optimizer in zip(self.optimizer_names, self.optimizers):
            state = torch.load(os.path.join(self._save_dir(), name + ".pth"), map_location=self.device)
            if state["name"] != name or state["epoch"] != epoch:
                raise RuntimeError("optimizer file %s.pth holds (%s, epoch %s), expected (%s, epoch %s)"
                                   % (name, state["name"], state["epoch"], name, epoch))
            optimizer.load_state_dict(state["state_dict"])
