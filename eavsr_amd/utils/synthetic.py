"""Deterministic synthetic weights and clips.

Trained checkpoints are not available (/root/reference/.MISSING_LARGE_BLOBS), so parity
and benchmarks run on seeded random weights.  The fill is a pure function of
(key, shape, preset, seed) -- it does not depend on module construction order -- so the
golden generator (which fills the *reference* modules), the oracle tests and the HIP
tests all see bit-identical parameters on any box with the same torch build.
"""
from __future__ import annotations

import hashlib
import math
from typing import Dict, Iterable, Tuple

import torch

FIXED_SUFFIXES = ("mean", "std", "regular_matrix")


def _gen_for(key: str, seed: int) -> torch.Generator:
    h = hashlib.sha256(f"{seed}:{key}".encode()).digest()
    g = torch.Generator(device="cpu")
    g.manual_seed(int.from_bytes(h[:7], "little"))
    return g


def _uniform(shape, bound: float, gen: torch.Generator) -> torch.Tensor:
    return (torch.rand(tuple(shape), generator=gen, dtype=torch.float32) * 2.0 - 1.0) * bound


def fill_state_dict(shapes: Dict[str, Tuple[int, ...]], preset: str = "default", seed: int = 0,
                    fixed: Dict[str, torch.Tensor] | None = None) -> Dict[str, torch.Tensor]:
    """Return {key: fp32 tensor}.

    preset 'default'      : PyTorch-default-like U(+-1/sqrt(fan_in)) weights and biases.
    preset 'trained_like' : same, but every ``transform_matrix_conv.bias`` is the identity
        [1,0,0,1] per deformable group plus a small perturbation, ``translation_conv.bias``
        is U(+-1.5) px, the residual-block convs use gain 1.5 and the encoder convs gain 2
        (features of O(1) instead of O(0.02)) -- so DCN taps do not
        collapse onto the centre pixel (SURVEY.md section 7 "synthetic-weight pathology")
        and the residual branches carry signal.
    Keys ending in mean/std/regular_matrix are constants of the architecture and are taken
    from ``fixed`` (or skipped when absent).
    """
    if preset not in ("default", "trained_like"):
        raise ValueError(preset)
    out: Dict[str, torch.Tensor] = {}
    for key in sorted(shapes):
        shape = tuple(shapes[key])
        leaf = key.rsplit(".", 1)[-1]
        if leaf in FIXED_SUFFIXES:
            if fixed is not None and key in fixed:
                out[key] = fixed[key].detach().clone().float()
            continue
        gen = _gen_for(key, seed)
        if leaf == "weight":
            fan_in = max(1, int(math.prod(shape[1:])))
            gain = 1.0
            if preset == "trained_like":
                if ".rg." in key and ".res." in key:
                    gain = 1.5
                elif key.startswith("encoder.") or ".encoder." in key:
                    gain = 2.0
            out[key] = _uniform(shape, gain / math.sqrt(fan_in), gen)
        elif leaf == "bias":
            wkey = key[:-4] + "weight"
            fan_in = max(1, int(math.prod(tuple(shapes[wkey])[1:]))) if wkey in shapes else shape[0]
            b = _uniform(shape, 1.0 / math.sqrt(fan_in), gen)
            if preset == "trained_like":
                if key.endswith("transform_matrix_conv.bias"):
                    ident = torch.tensor([1.0, 0.0, 0.0, 1.0]).repeat(shape[0] // 4)
                    b = ident + _uniform(shape, 0.25, gen)
                elif key.endswith("translation_conv.bias"):
                    b = _uniform(shape, 1.5, gen)
            out[key] = b
        else:
            out[key] = _uniform(shape, 1.0, gen)
    return out


def synthetic_clip(n: int, t: int, h: int, w: int, seed: int = 0, smooth: bool = True) -> torch.Tensor:
    """(n,t,3,h,w) fp32 in [0,1).  ``smooth=False`` is plain torch.rand.  With
    ``smooth=True`` the frames are a low-pass random texture translated by a per-frame
    sub-pixel drift plus a little noise, so SPyNet/DCN see coherent motion instead of
    white noise (white noise makes every sampler read uncorrelated pixels)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(1000 + seed)
    if not smooth:
        return torch.rand(n, t, 3, h, w, generator=g)
    base = torch.rand(n, 3, h // 4 + 8, w // 4 + 8, generator=g)
    base = torch.nn.functional.interpolate(base, scale_factor=4, mode="bicubic", align_corners=False)
    frames = []
    for i in range(t):
        dy = 6 + int(round(2.0 * math.sin(0.9 * i)))
        dx = 6 + i
        f = base[:, :, dy:dy + h, dx:dx + w]
        frames.append(f + 0.03 * torch.rand(n, 3, h, w, generator=g))
    return torch.stack(frames, 1).clamp(0.0, 0.999).contiguous()


def shapes_of(sd: Dict[str, torch.Tensor]) -> Dict[str, Tuple[int, ...]]:
    return {k: tuple(v.shape) for k, v in sd.items()}


def keys_digest(shapes: Dict[str, Iterable[int]]) -> str:
    s = ";".join(f"{k}:{','.join(map(str, shapes[k]))}" for k in sorted(shapes))
    return hashlib.sha256(s.encode()).hexdigest()
