"""Evaluation harness either side of the hot path (f4).

Counterparts of the reference's test loop (test_basic.py:56-83: set_input -> synchronize -> model.test() ->
synchronize, PSNR on the clamp*255*round visuals), of `calc_psnr` (util/util.py:302-320) and of the frame-window
index maps of the datasets (data/mvsr4x_dataset.py:105-147), of the frame writing of test_basic.py:85-92 (8-bit RGB PNG
files under `<root>/sr_{full,patch}_<iter>/<scene>/<frame>`; a dependency-free encoder: zlib + struct) and of the SSIM of
psnr_total.py:39-44 (skimage's `structural_similarity(win_size=11, data_range=255, multichannel=True,
gaussian_weights=True)`, restated in torch).  LPIPS (psnr_total.py:27-35) needs the pretrained AlexNet blob of the `lpips`
package, which does not exist here: not provided.  No dataset files: the benchmarks feed synthetic clips, a user's loader
feeds `{'lr_seq', 'hr_seq', 'fname'}` dicts exactly as the reference's does.
"""
from __future__ import annotations

import math
import os
import struct
import time
import zlib
from typing import Dict, Iterable, List, Optional, Sequence

import torch

Tensor = torch.Tensor


def calc_psnr(sr: Tensor, hr: Tensor, range: float = 255.0) -> float:
    """-10 log10(mean(((sr - hr) / range)^2)) over the whole tensor (util/util.py:302-320); inputs are the
    `get_current_visuals()` tensors (already x255, clamped and rounded)."""
    with torch.no_grad():
        diff = (sr.float() - hr.float()) / range
        mse = torch.pow(diff, 2).mean()
        return (-10 * torch.log10(mse)).item()


def calc_ssim(sr: Tensor, hr: Tensor, data_range: float = 255.0, win_size: int = 11, sigma: float = 1.5) -> float:
    """Mean SSIM of two images as psnr_total.py:39-44 computes it: skimage.metrics.structural_similarity(out, ref,
    win_size=11, data_range=255, multichannel=True, gaussian_weights=True) -- per channel, an 11-tap gaussian window of
    sigma 1.5 (truncate 3.5), sample covariances (x NP / (NP - 1), NP = 11^2), K1 = 0.01, K2 = 0.03, the SSIM map cropped by
    (win_size - 1) / 2 on every side, mean over pixels and channels; float64 arithmetic as skimage's for 8-bit inputs.
    sr / hr: (..., C, H, W) tensors in [0, data_range] (the `get_current_visuals()` frames); leading dimensions are averaged."""
    if sr.shape != hr.shape or sr.dim() < 3:
        raise ValueError(f"calc_ssim: shapes {tuple(sr.shape)} / {tuple(hr.shape)}")
    h, w = sr.shape[-2:]
    if min(h, w) < win_size:
        raise ValueError(f"calc_ssim: image {h} x {w} smaller than the {win_size}-tap window")
    with torch.no_grad():
        x = sr.reshape(-1, 1, h, w).to(torch.float64)
        y = hr.reshape(-1, 1, h, w).to(torch.float64)
        r = win_size // 2
        k = torch.exp(-0.5 * (torch.arange(-r, r + 1, dtype=torch.float64, device=x.device) / sigma) ** 2)
        k = k / k.sum()
        kh, kw = k.view(1, 1, -1, 1), k.view(1, 1, 1, -1)
        filt = lambda t: torch.nn.functional.conv2d(torch.nn.functional.conv2d(t, kh), kw)      # "valid": exactly the cropped region
        ux, uy = filt(x), filt(y)
        cov_norm = (win_size * win_size) / (win_size * win_size - 1.0)
        vx = cov_norm * (filt(x * x) - ux * ux)
        vy = cov_norm * (filt(y * y) - uy * uy)
        vxy = cov_norm * (filt(x * y) - ux * uy)
        c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
        s_map = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
        return s_map.mean().item()


def _png_chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def write_png(image: Tensor, path: str, level: int = 6) -> str:
    """One 8-bit image, (3, H, W) RGB or (1, H, W) / (H, W) grey, values 0..255 (a `get_current_visuals()` frame: already
    clamped and rounded), as a PNG file -- what `dataset_test.imio.write(np.array(frame).astype(np.uint8), path)` leaves
    (test_basic.py:85-92; data/imlib.py:164-166 creates the directory).  Colour type 2 / 0, bit depth 8, filter 0, one IDAT."""
    t = image.detach()
    if t.dim() == 2:
        t = t.unsqueeze(0)
    if t.dim() != 3 or t.shape[0] not in (1, 3):
        raise ValueError(f"write_png: (3, H, W), (1, H, W) or (H, W), got {tuple(image.shape)}")
    c, h, w = (int(v) for v in t.shape)
    u8 = t.to(torch.float32).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).contiguous().cpu()      # astype(np.uint8): truncation
    rows = torch.cat([torch.zeros(h, 1, dtype=torch.uint8), u8.view(h, w * c)], 1)                  # filter byte 0 per scanline
    raw = rows.numpy().tobytes()
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(_png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2 if c == 3 else 0, 0, 0, 0)))
        f.write(_png_chunk(b"IDAT", zlib.compress(raw, level)))
        f.write(_png_chunk(b"IEND", b""))
    return path


def read_png(path: str) -> Tensor:
    """Decoder for the files `write_png` writes and for any non-interlaced 8-bit grey / RGB / RGBA PNG (all five scanline
    filters): (C, H, W) uint8.  For round trips and for feeding stored frames back; not a general PNG reader."""
    data = open(path, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if zlib.crc32(tag + body) & 0xFFFFFFFF != struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]:
            raise ValueError(f"{path}: CRC mismatch in chunk {tag!r}")
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat.append(body)
        pos += 12 + n
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or ctype not in (0, 2, 6) or interlace:
        raise ValueError(f"{path}: only non-interlaced 8-bit grey / RGB / RGBA")
    c = {0: 1, 2: 3, 6: 4}[ctype]
    raw = bytearray(zlib.decompress(b"".join(idat)))
    stride = w * c
    out = bytearray(h * stride)
    prev = bytearray(stride)
    for y in range(h):
        ft = raw[y * (stride + 1)]
        line = raw[y * (stride + 1) + 1:(y + 1) * (stride + 1)]
        if ft == 1:
            for i in range(c, stride):
                line[i] = (line[i] + line[i - c]) & 255
        elif ft == 2:
            for i in range(stride):
                line[i] = (line[i] + prev[i]) & 255
        elif ft == 3:
            for i in range(stride):
                line[i] = (line[i] + (((line[i - c] if i >= c else 0) + prev[i]) >> 1)) & 255
        elif ft == 4:
            for i in range(stride):
                a = line[i - c] if i >= c else 0
                b = prev[i]
                cc = prev[i - c] if i >= c else 0
                pa, pb, pc = abs(b - cc), abs(a - cc), abs(a + b - 2 * cc)
                line[i] = (line[i] + (a if (pa <= pb and pa <= pc) else b if pb <= pc else cc)) & 255
        elif ft != 0:
            raise ValueError(f"{path}: scanline filter {ft}")
        out[y * stride:(y + 1) * stride] = line
        prev = line
    return torch.frombuffer(out, dtype=torch.uint8).view(h, w, c).permute(2, 0, 1).contiguous()


def save_visuals(res: Dict[str, Tensor], fnames: Sequence, root: str, load_iter="0", full_res: bool = False) -> List[str]:
    """The frame writing of test_basic.py:85-92 for one test item: frame i of `res['data_sr_seq'][0]` goes to
    `<root>/sr_{full|patch}_<load_iter>/<fname[i][0][:3]>/<fname[i][0][-9:]>` (the scene is the first three characters of the
    frame's file name, the file its last nine: `000/00000.png`-style names of the datasets).  `root` is the reference's
    `./ckpt/<opt.name>`; `fnames` the item's `data['fname']` (a list of one-element lists / tuples, as the DataLoader collates
    them, or plain strings).  Returns the written paths."""
    seq = res["data_sr_seq"]
    if seq.dim() != 5:
        raise ValueError(f"save_visuals: data_sr_seq must be (n, t, c, h, w), got {tuple(seq.shape)}")
    paths = []
    for i in range(seq.shape[1]):
        name = fnames[i]
        name = name[0] if isinstance(name, (list, tuple)) else name
        folder = os.path.join(root, "sr_%s_%s" % ("full" if full_res else "patch", load_iter), name[:3])
        paths.append(write_png(seq[0, i], os.path.join(folder, name[-9:])))
    return paths


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


def evaluate(model, items: Iterable[Dict], calc_psnr_flag: bool = True, calc_ssim_flag: bool = False,
             save_root: Optional[str] = None, load_iter="0", full_res: bool = False) -> Dict:
    """The timed loop of test_basic.py:56-83 for a model wrapper (EAVSRPModel / EAVSRPx2Model); with `save_root` the frames are
    written as the reference's `--save_imgs` does (test_basic.py:85-92, `save_visuals`); `calc_ssim_flag` adds psnr_total.py's SSIM
    per item (mean over the item's frames).

    Every item is a `{'lr_seq': (n,t,3,h,w), 'hr_seq': (n,t,3,sh,sw), 'fname': ...}` dict in [0,1].  Returns the
    per-item PSNR list, their mean, the wall time of the `model.test()` calls (device-synchronised on both sides,
    as the reference's) and frames/s over all items (the reference discards its first iteration inside
    `EAVSRPModel.forward`, eavsrp_model.py:104-107; `model.time` / `model.num` keep that convention)."""
    model.eval()
    psnr: List[float] = []
    ssim: List[float] = []
    written: List[str] = []
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
        res = None
        if (calc_psnr_flag or calc_ssim_flag) and model.data_hr_seq is not None:
            res = model.get_current_visuals()
            if calc_psnr_flag:
                psnr.append(calc_psnr(res["data_sr_seq"], res["data_hr_seq"]))
            if calc_ssim_flag:
                ssim.append(calc_ssim(res["data_sr_seq"], res["data_hr_seq"]))
        if save_root is not None:
            res = res if res is not None else model.get_current_visuals()
            written += save_visuals(res, data["fname"], save_root, load_iter, full_res)
    return {
        "psnr": psnr,
        "psnr_mean": (sum(psnr) / len(psnr)) if psnr else math.nan,
        "ssim": ssim,
        "ssim_mean": (sum(ssim) / len(ssim)) if ssim else math.nan,
        "written": written,
        "seconds": seconds,
        "frames": frames,
        "frames_per_s": frames / seconds if seconds > 0 else math.nan,
    }
