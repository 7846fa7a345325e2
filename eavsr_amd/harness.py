"""Evaluation harness either side of the hot path (f4).

Counterparts of the reference's test loop (test_basic.py:56-83: set_input -> synchronize -> model.test() ->
synchronize, PSNR on the clamp*255*round visuals), of `calc_psnr` (util/util.py:302-320) and of the frame-window
index maps of the datasets (data/mvsr4x_dataset.py:105-147).  No image IO, no dataset files: the benchmarks feed
synthetic clips, a user's loader feeds `{'lr_seq', 'hr_seq', 'fname'}` dicts exactly as the reference's does.
"""
from __future__ import annotations

import math
import time
from typing import Dict, Iterable, List, Sequence

import torch

Tensor = torch.Tensor


def calc_psnr(sr: Tensor, hr: Tensor, range: float = 255.0) -> float:
    """-10 log10(mean(((sr - hr) / range)^2)) over the whole tensor (util/util.py:302-320); inputs are the
    `get_current_visuals()` tensors (already x255, clamped and rounded)."""
    with torch.no_grad():
        diff = (sr.float() - hr.float()) / range
        mse = torch.pow(diff, 2).mean()
        return (-10 * torch.log10(mse)).item()


def test_window_starts(n_images: int, n_seq: int, n_frame: int) -> List[int]:
    """First frame of every test item (mvsr4x_dataset.py:130-136): each scene of `n_seq` consecutive frames is cut
    into n_seq / n_frame non-overlapping windows of n_frame frames.  Raises as the reference does when n_seq is not
    a multiple of n_frame."""
    if n_seq % n_frame != 0:
        raise ValueError(f"n_seq {n_seq} is not a multiple of n_frame {n_frame}")
    per_scene = n_seq // n_frame
    index = [i * n_frame for i in range(per_scene)]
    n_items = (n_images // n_seq) * per_scene
    return [(i // per_scene) * n_seq + index[i % per_scene] for i in range(n_items)]


def train_window(idx: int, frame: int, n_frame: int, n_seq: int) -> List[int]:
    """Image indices of the training / validation item whose key frame is image `idx`, the `frame`-th frame of its
    scene of `n_seq` frames (mvsr4x_dataset.py:62-90, :97-125): a window of n_frame frames centred on the key frame;
    at the first / last frames of a scene the missing neighbours are mirrored about the key frame, so the window
    never crosses into another scene."""
    half = n_frame // 2
    out = [0] * n_frame
    if frame - half < 0:                      # front of the scene
        for i in range(half - frame):
            out[i] = idx + half - i
        for i in range(half - frame, n_frame):
            out[i] = idx + i - half
    elif frame + half >= n_seq:               # back of the scene
        for i in range(half, (n_seq - 1) - frame, -1):
            out[i + half] = idx - i
        for i in range(half + n_seq - frame):
            out[i] = idx + i - half
    else:
        for i in range(n_frame):
            out[i] = idx + i - half
    return out


def crop_center(img: Tensor, p: int) -> Tensor:
    """Centre p x p crop of an (..., H, W) tensor (the `_crop_center` of the training items, mvsr4x_dataset.py:123-124)."""
    h, w = img.shape[-2:]
    top, left = (h - p) // 2, (w - p) // 2
    return img[..., top:top + p, left:left + p]


def evaluate(model, items: Iterable[Dict], calc_psnr_flag: bool = True) -> Dict:
    """The timed loop of test_basic.py:56-83 for a model wrapper (EAVSRPModel / EAVSRPx2Model).

    Every item is a `{'lr_seq': (n,t,3,h,w), 'hr_seq': (n,t,3,sh,sw), 'fname': ...}` dict in [0,1].  Returns the
    per-item PSNR list, their mean, the wall time of the `model.test()` calls (device-synchronised on both sides,
    as the reference's) and frames/s over all items (the reference discards its first iteration inside
    `EAVSRPModel.forward`, eavsrp_model.py:104-107; `model.time` / `model.num` keep that convention)."""
    model.eval()
    psnr: List[float] = []
    seconds = 0.0
    frames = 0
    for data in items:
        model.set_input(data, 0)
        torch.cuda.synchronize()
        t0 = time.time()
        model.test()
        torch.cuda.synchronize()
        seconds += time.time() - t0
        frames += int(model.data_sr_seq.shape[0] * model.data_sr_seq.shape[1])
        if calc_psnr_flag and model.data_hr_seq is not None:
            res = model.get_current_visuals()
            psnr.append(calc_psnr(res["data_sr_seq"], res["data_hr_seq"]))
    return {
        "psnr": psnr,
        "psnr_mean": (sum(psnr) / len(psnr)) if psnr else math.nan,
        "seconds": seconds,
        "frames": frames,
        "frames_per_s": frames / seconds if seconds > 0 else math.nan,
    }
