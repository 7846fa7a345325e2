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


# G4 / G9 ----------------------------------------------------------------------------
# DCNv2 known answers: (n, c, h, w, cout, dg, sigma, seed).  c = cout = 64, dg = 8 is the reference's configuration
# (eavsrp_model.py:143); sigma = 8 px on a 12 x 16 image puts most samples across or beyond the border, the 'edge' case
# pins offsets to the exact validity boundary p in {-1, -1 + eps, size - eps, size} of the `-1 < p < size` gate.
G4_CASES = {
    "dg8_s0p5": (1, 64, 12, 16, 64, 8, 0.5, 401),
    "dg8_s2": (2, 64, 12, 16, 64, 8, 2.0, 402),
    "dg8_s8": (1, 64, 12, 16, 64, 8, 8.0, 403),
    "dg8_edge": (1, 64, 8, 12, 64, 8, -1.0, 404),
    "dg2_s2_c16": (1, 16, 9, 11, 8, 2, 2.0, 405),
    "dg1_s1p5_ragged": (1, 8, 7, 5, 4, 1, 1.5, 406),
}


def g4_inputs(name):
    n, c, h, w, co, dg, sigma, seed = G4_CASES[name]
    x = randn(seed, n, c, h, w)
    if sigma >= 0:
        off = randn(seed + 1, n, dg * 18, h, w, scale=sigma)
    else:
        # every tap's sampling position lands on / next to the validity boundary of the image, by construction:
        # p = y - 1 + i + dy  ->  dy = target - (y - 1 + i), target cycling through the boundary values
        eps = 2.0 ** -10
        ty = torch.tensor([-1.0, -1.0 + eps, -0.5, h - 1.0, h - eps, float(h), h - 0.5, 0.0])
        tx = torch.tensor([-1.0, -1.0 + eps, -0.5, w - 1.0, w - eps, float(w), w - 0.5, 0.0])
        off = torch.zeros(n, dg, 9, 2, h, w)
        ys = torch.arange(h, dtype=torch.float32).view(h, 1)
        xs = torch.arange(w, dtype=torch.float32).view(1, w)
        pick = torch.randint(0, 8, (n, dg, 9, 2, h, w), generator=_g(seed + 1))
        for k in range(9):
            i, j = k // 3, k % 3
            off[:, :, k, 0] = ty[pick[:, :, k, 0]] - (ys - 1 + i)
            off[:, :, k, 1] = tx[pick[:, :, k, 1]] - (xs - 1 + j)
        off = off.view(n, dg * 18, h, w)
    mask = rand(seed + 2, n, dg * 9, h, w)
    wt = randn(seed + 3, co, c, 3, 3, scale=1.0 / (3.0 * c ** 0.5))
    b = randn(seed + 4, co, scale=0.1)
    return x, off, mask, wt, b, dg


def g9_cotangent(name, shape):
    """the seeded upstream gradient of the G9 gradient fixtures"""
    return randn(900 + sum(map(ord, name)), *shape)
