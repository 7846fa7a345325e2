#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference modules.

Runs only in the build container (needs /root/reference, read-only).  Nothing here
travels: the outputs are data (.npz / .json), this script is the recipe.

The reference cannot be imported as-is: torchvision, mmcv, cv2 and cupy are absent from the
image.  We install sys.modules stand-ins for exactly those
third-party packages (no reference file is modified or copied):
  cv2, cupy                      empty modules (cupy.memoize -> no-op decorator)
  torchvision.models.vgg.vgg16   randomly initialised VGG16 `features` Sequential
  torchvision.ops                empty
  mmcv.utils.get_logger          logging.getLogger
  mmcv.runner.load_checkpoint    raises (never called: pretrained=None)
  mmcv.cnn.ConvModule            Conv2d + optional ReLU with mmcv's attribute names
                                 (.conv, .activate) so state_dict keys match
  mmcv.ops.ModulatedDeformConv2d / modulated_deform_conv2d
                                 parameter container + *our oracle's* dcnv2 restatement.
DCNv2 is therefore the one piece whose arithmetic is NOT the reference's own
(mmcv is not in /root/reference): parity for it is unpinned and anchored on known-answer
tests instead (tests/test_oracle_dcn.py).

Usage:  python tests/golden/gen_golden.py            (about a minute on 8 cores)
"""
from __future__ import annotations

import json
import logging
import math
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from oracle import eavsr_oracle as O  # noqa: E402  (only for the mmcv DCNv2 stand-in)
from eavsr_amd.utils.synthetic import fill_state_dict, shapes_of, keys_digest  # noqa: E402
from tests.golden import cases  # noqa: E402


# ----------------------------------------------------------------------------------
# third-party stand-ins
# ----------------------------------------------------------------------------------
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_shims():
    _mod("cv2")
    cupy = _mod("cupy")
    cupy.memoize = lambda **kw: (lambda f: f)
    _mod("tensorboardX", SummaryWriter=object)

    def _vgg16(pretrained=False, **kw):
        cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
        layers, cin = [], 3
        for v in cfg:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        return Namespace(features=nn.Sequential(*layers))

    tv = _mod("torchvision")
    tv.models = _mod("torchvision.models")
    tv.models.vgg = _mod("torchvision.models.vgg", vgg16=_vgg16)
    tv.ops = _mod("torchvision.ops")
    tv.models.vgg16 = _vgg16

    mmcv = _mod("mmcv")
    mmcv.utils = _mod("mmcv.utils", get_logger=lambda name, **kw: logging.getLogger(name))

    def _load_checkpoint(*a, **k):
        raise RuntimeError("checkpoint blobs are absent (.MISSING_LARGE_BLOBS)")

    mmcv.runner = _mod("mmcv.runner", load_checkpoint=_load_checkpoint)

    class ConvModule(nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                     norm_cfg=None, act_cfg=dict(type="ReLU"), **kw):
            super().__init__()
            assert norm_cfg is None
            self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding)
            self.with_activation = act_cfg is not None
            if self.with_activation:
                assert act_cfg["type"] == "ReLU"
                self.activate = nn.ReLU(inplace=True)

        def forward(self, x):
            x = self.conv(x)
            return self.activate(x) if self.with_activation else x

    mmcv.cnn = _mod("mmcv.cnn", ConvModule=ConvModule)

    def _one(v):
        return v[0] if isinstance(v, (tuple, list)) else v

    def modulated_deform_conv2d(input, offset, mask, weight, bias, stride, padding, dilation,
                                groups, deform_groups):
        return O.dcnv2(input, offset, mask, weight, bias, _one(stride), _one(padding),
                       _one(dilation), groups, deform_groups)

    class ModulatedDeformConv2d(nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                     dilation=1, groups=1, deform_groups=1, bias=True):
            super().__init__()
            ks = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
            self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, ks
            self.stride, self.padding, self.dilation = (stride, stride), (padding, padding), (dilation, dilation)
            self.groups, self.deform_groups = groups, deform_groups
            self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *ks))
            self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
            stdv = 1.0 / math.sqrt(in_channels * ks[0] * ks[1])
            self.weight.data.uniform_(-stdv, stdv)

        def forward(self, x, offset, mask):
            return modulated_deform_conv2d(x, offset, mask, self.weight, self.bias, self.stride,
                                           self.padding, self.dilation, self.groups, self.deform_groups)

    mmcv.ops = _mod("mmcv.ops", ModulatedDeformConv2d=ModulatedDeformConv2d,
                    modulated_deform_conv2d=modulated_deform_conv2d)


def load_filled(module: nn.Module, preset: str, prefix: str = ""):
    """Overwrite every parameter of a reference module with the deterministic fill, keyed by
    `prefix + state_dict key` (the same keys the tests use)."""
    sd = module.state_dict()
    shapes = {prefix + k: tuple(v.shape) for k, v in sd.items()}
    filled = fill_state_dict(shapes, preset, fixed={prefix + k: v for k, v in sd.items()})
    module.load_state_dict({k[len(prefix):]: v for k, v in filled.items()}, strict=True)
    return filled


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1e3:.0f} kB)")


def main():
    torch.set_num_threads(os.cpu_count())
    install_shims()
    from models import networks as RN            # noqa: the reference, read-only
    from models import eavsrp_model as RM        # noqa
    from models import eavsrpx2_model as RM2     # noqa

    opt = Namespace(predict=False, n_frame=7, n_flow=5, scale=4)
    torch.manual_seed(0)

    with torch.no_grad():
        # ---- G1 flow_warp (networks.py:699-739 NCHW flow; eavsrp_model.py:587-626 NHWC flow)
        out = {}
        for name, (x, flow, pad) in cases.g1_flow_warp_cases().items():
            out[name + "__nchw"] = RN.flow_warp(x, flow, padding_mode=pad)
            out[name + "__nhwc"] = RM.flow_warp(x, flow.permute(0, 2, 3, 1).contiguous(), padding_mode=pad)
        save("g1_flow_warp", **out)

        # ---- G2 AdaptBlock2_3x3 + TransOffsetworelu
        blk = RN.AdaptBlock2_3x3(opt, 64, 64, deformable_groups=8).eval()
        tr = RN.TransOffsetworelu().eval()
        for preset in ("default", "trained_like"):
            load_filled(blk, preset, "g2.flow.")
            load_filled(tr, preset, "g2.trans.")
            x, h = cases.g2_inputs()
            off = blk(x, h)
            out = {"offset18": off, "flow2": tr(off)}
            save(f"g2_adapt3x3_{preset}", **out)

        # ---- G3 AdaptBlockOffset (D=8)
        blk = RN.AdaptBlockOffset(opt, 64, 64, deformable_groups=8).eval()
        for preset in ("default", "trained_like"):
            load_filled(blk, preset, "g3.adastn.")
            x, h = cases.g3_inputs()
            off, mask = blk(x, h)
            save(f"g3_adaptoffset_{preset}", offset=off, mask=mask)

        # ---- G5 MultiAdSTN (DCNv2 step = stand-in, see header)
        m = RN.MultiAdSTN(opt, 64, 64, deformable_groups=8).eval()
        for preset in ("default", "trained_like"):
            load_filled(m, preset, "g5.align.")
            nbr, ref, fp, flow = cases.g5_inputs()
            save(f"g5_multiadstn_{preset}", out=m(nbr, ref, fp, flow))

        # ---- G6 RCAB / RCAGroup / ResidualBlocksWithInputConv
        x64, x128 = cases.g6_inputs()
        for preset in ("default", "trained_like"):
            a = RN.RCABlock(64, 64).eval()
            load_filled(a, preset, "g6.rcab.")
            b = RN.RCAGroup(64, 64, nb=2).eval()
            load_filled(b, preset, "g6.group.")
            c = RM.ResidualBlocksWithInputConv(128, 64, 2).eval()
            load_filled(c, preset, "g6.rbic.")
            save(f"g6_backbone_{preset}", rcab=a(x64), group=b(x64), rbic=c(x128))

        # ---- G7 propagate + G8 end-to-end, x4 model
        for scale, mod, tag in ((4, RM, "x4"), (2, RM2, "x2")):
            o = Namespace(predict=False, n_frame=7, n_flow=5, scale=scale)
            cls = mod.EAVSRP if scale == 4 else mod.EAVSRPx2 if hasattr(mod, "EAVSRPx2") else None
            if cls is None:
                cls = [v for k, v in vars(mod).items() if k.startswith("EAVSRP") and isinstance(v, type)
                       and issubclass(v, nn.Module)][0]
            net = cls(o, None).eval()
            shapes = shapes_of(net.state_dict())
            with open(os.path.join(HERE, f"eavsrp_{tag}_keys.json"), "w") as f:
                json.dump({"class": cls.__name__, "digest": keys_digest(shapes),
                           "n_tensors": len(shapes),
                           "n_params": int(sum(math.prod(s) for s in shapes.values())),
                           "shapes": {k: list(v) for k, v in sorted(shapes.items())}}, f, indent=0)
            for preset in ("default", "trained_like"):
                load_filled(net, preset)
                if scale == 4:
                    feats, flows, prev = cases.g7_inputs()
                    f_b1 = {k: list(v) for k, v in feats.items()}
                    f_b1["backward_1"] = []
                    r1 = net.propagate(f_b1, flows, "backward_1")["backward_1"]
                    f_f2 = {k: list(v) for k, v in feats.items()}
                    for k in ("backward_1", "forward_1", "backward_2"):
                        f_f2[k] = list(prev[k])
                    f_f2["forward_2"] = []
                    r2 = net.propagate(f_f2, flows, "forward_2")["forward_2"]
                    save(f"g7_propagate_{preset}", backward_1=torch.stack(r1, 1), forward_2=torch.stack(r2, 1))
                clip = cases.g8_clip()
                ff, fb = net.compute_flow(clip)
                caps = []   # inputs of `reconstruction` = cat(spatial, 4 branch features) per frame
                hook = net.reconstruction.register_forward_pre_hook(lambda m_, inp: caps.append(inp[0]))
                y = net(clip)
                hook.remove()
                save(f"g8_e2e_{tag}_{preset}", sub=cases.subsample(y),
                     branch_feats=torch.stack([cases.subsample(caps[i]) for i in (0, 3, 6)], 1),
                     mean=y.mean(dim=(-1, -2)), absmean=y.abs().mean(dim=(-1, -2)),
                     flows_forward=ff, flows_backward=fb,
                     shape=np.array(y.shape))
                print(tag, preset, "out range", float(y.min()), float(y.max()))


if __name__ == "__main__":
    main()
