"""Seeded input generators shared by gen_golden.py (which feeds them to the REFERENCE
modules in the build container) and by the tests (which feed them to the oracle / HIP
path).  Inputs are regenerated from seeds instead of being stored, outputs are stored
in the .npz fixtures next to this file.  torch's CPU RNG stream is fixed for the torch
build pinned in this image."""
from __future__ import annotations

import torch


def _g(seed):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return g


def randn(seed, *shape, scale=1.0):
    return torch.randn(*shape, generator=_g(seed)) * scale


def rand(seed, *shape):
    return torch.rand(*shape, generator=_g(seed))


# G1 -------------------------------------------------------------------------------
def g1_flow_warp_cases():
    """name -> (x, flow_nchw, padding_mode).  Flows include far out-of-range samples."""
    cases = {}
    for name, (n, c, h, w, sigma, pad, seed) in {
        "c2_zeros": (2, 2, 13, 17, 2.0, "zeros", 11),
        "c3_border": (2, 3, 12, 20, 3.0, "border", 12),
        "c64_zeros": (1, 64, 16, 24, 2.0, "zeros", 13),
        "c5_zeros_far": (1, 5, 9, 11, 12.0, "zeros", 14),
        "c3_border_far": (1, 3, 9, 11, 12.0, "border", 15),
        "c4_int_zeros": (1, 4, 8, 8, 0.0, "zeros", 16),
    }.items():
        x = randn(seed, n, c, h, w)
        flow = randn(seed + 100, n, 2, h, w, scale=sigma)
        if sigma == 0.0:  # exact-integer displacements incl. the image edge
            flow = torch.randint(-3, 4, (n, 2, h, w), generator=_g(seed + 200)).float()
        cases[name] = (x, flow, pad)
    return cases


# G2 / G3 --------------------------------------------------------------------------
def g2_inputs():
    return randn(21, 1, 64, 12, 16), randn(22, 1, 64, 12, 16)


def g3_inputs():
    return randn(31, 2, 64, 10, 12), randn(32, 2, 64, 10, 12)


# G5 -------------------------------------------------------------------------------
def g5_inputs(h=32, w=48, n=1):
    nbr = [randn(51 + i, n, 64, h >> i, w >> i) for i in range(3)]
    ref = [randn(54 + i, n, 64, h >> i, w >> i) for i in range(3)]
    feat_prop = randn(57, n, 64, h, w)
    flow = randn(58, n, 2, h, w, scale=1.5)
    return nbr, ref, feat_prop, flow


# G6 -------------------------------------------------------------------------------
def g6_inputs():
    return randn(61, 2, 64, 20, 24), randn(62, 2, 128, 20, 24)


# G7 -------------------------------------------------------------------------------
def g7_inputs(h=32, w=32, n=1, t=3):
    feats = {
        "spatial": [randn(700 + i, n, 64, h, w) for i in range(t)],
        "spatial_d2": [randn(710 + i, n, 64, h // 2, w // 2) for i in range(t)],
        "spatial_d4": [randn(720 + i, n, 64, h // 4, w // 4) for i in range(t)],
    }
    flows = randn(730, n, t - 1, 2, h, w, scale=1.5)
    prev = {
        "backward_1": [randn(740 + i, n, 64, h, w) for i in range(t)],
        "forward_1": [randn(750 + i, n, 64, h, w) for i in range(t)],
        "backward_2": [randn(760 + i, n, 64, h, w) for i in range(t)],
    }
    return feats, flows, prev


# G8 -------------------------------------------------------------------------------
def g8_clip(n=1, t=7, h=64, w=64):
    from eavsr_amd.utils.synthetic import synthetic_clip
    return synthetic_clip(n, t, h, w, seed=8)


def subsample(out):
    """The part of an end-to-end output kept in the fixture."""
    return out[..., 1::4, 2::4].contiguous()
