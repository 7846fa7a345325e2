"""Functional host-side wrappers over the C ABI (include/eavsr_hip.h).

PyTorch is plumbing here: it owns device memory (outputs come from torch.empty on the caching
allocator) and the stream (torch.cuda.current_stream()).  Every function launches HIP kernels
from libeavsr_hip.so asynchronously on that stream and raises on any error.  Inputs must be
fp32 CUDA(HIP) tensors; there is no CPU path.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import weakref
from typing import List, Optional, Sequence, Tuple, Union

import torch

from . import _native as N

Tensor = torch.Tensor

ACT = {"none": 0, None: 0, "relu": 1, "lrelu": 2, "relu_mask": 3}      # relu_mask: conv2d(residual = the ReLU's forward output), direct kernels only


def _chk(t: Tensor, name: str) -> Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor is on {t.device}; eavsr_amd runs on the GPU only (no CPU path)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: dtype {t.dtype} unsupported (fp32 only in this round)")
    return t if t.is_contiguous() else t.contiguous()


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _stream(t: Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream


class _DeviceOf:
    """Make sure launches go to the device that owns the tensors."""

    def __init__(self, t: Tensor):
        self.idx = t.device.index
        self.prev = None

    def __enter__(self):
        cur = torch.cuda.current_device()
        if cur != self.idx:
            self.prev = cur
            torch.cuda.set_device(self.idx)

    def __exit__(self, *a):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)


def lib():
    return N.load()


class LabBuildRequired(RuntimeError):
    """A mode / op that exists only in the lab build of the library (`python -m eavsr_amd.build --lab`): the retired schedules
    kept for A/B measurements (eavsr_amd/build.py LAB_SOURCES; the EXPERIMENTAL section of include/eavsr_hip.h)."""


def lab_available() -> bool:
    return N.lab_build()


def require_lab(what: str) -> None:
    if not lab_available():
        raise LabBuildRequired(f"{what} is part of the lab build only: rebuild with `python -m eavsr_amd.build --lab` "
                               "(the default library holds the product path and its documented modes)")


def _wino_tiles(h: int, w: int) -> int:
    """8 x 32-pixel tiles of an image (the granularity the Winograd gates count in; eavsr_conv3x3_wino_tiles of the lab build)"""
    return ((h + 7) // 8) * ((w + 31) // 32)


# ------------------------------------------------------------------------------------------
# optional per-launch timing (HIP events on the launch stream); off unless `with profile():`
# ------------------------------------------------------------------------------------------
class Profile:
    """Collects (kernel name, algorithmic flops, algorithmic bytes, start/end event) per launch.
    Events are recorded on the stream the kernel is launched on, so the elapsed time is the
    device-side duration of that launch (plus ~1 us of event overhead)."""

    def __init__(self):
        self.records = []

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, flops, nbytes, e0, e1 in self.records:
            d = out.setdefault(name, {"calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["calls"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["bytes"] += nbytes
        return out

    def by_flops(self, name: str):
        """the launches of one kernel name grouped by their algorithmic FLOP count (= by launch shape): {flops: (calls, ms)}"""
        torch.cuda.synchronize()
        out = {}
        for nm, flops, _nbytes, e0, e1 in self.records:
            if nm == name:
                c, ms = out.get(flops, (0, 0.0))
                out[flops] = (c + 1, ms + e0.elapsed_time(e1))
        return out


_prof: Optional[Profile] = None


class profile:
    def __enter__(self):
        global _prof
        _prof = Profile()
        return _prof

    def __exit__(self, *a):
        global _prof
        _prof = None


def _launch(name: str, flops: float, nbytes: float, t: Tensor, fn, what: str):
    """Run `fn()` (a C-ABI call returning an int status) on t's device and check it."""
    with _DeviceOf(t):
        if _prof is None:
            code = fn()
        else:
            st = torch.cuda.current_stream(t.device)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(st)
            code = fn()
            e1.record(st)
            _prof.records.append((name, flops, nbytes, e0, e1))
    N.check(code, what)


def selftest_mfma(device="cuda:0") -> int:
    """Number of accumulator elements that disagree with the assumed MFMA lane layout (0 = OK)."""
    s = torch.zeros(8192, device=device, dtype=torch.float32)
    with _DeviceOf(s):
        N.check(lib().eavsr_selftest_mfma_f32(s.data_ptr(), _stream(s)), "selftest_mfma")
    return int(s[0].item())


# ------------------------------------------------------------------------------------------
# flow_warp  (networks.py:699-739, eavsrp_model.py:587-626)
# ------------------------------------------------------------------------------------------
def flow_warp(x: Tensor, flow: Tensor, padding_mode: str = "zeros", flow2: Optional[Tensor] = None,
              flow_layout: str = "nchw", interpolation: str = "bilinear", align_corners: bool = True) -> Tensor:
    x = _chk(x, "x")
    n, c, h, w = x.shape
    if flow_layout == "nchw":
        if tuple(flow.shape) != (n, 2, h, w):
            raise ValueError(f"The spatial sizes of input ({(h, w)}) and flow ({tuple(flow.shape)}) are not the same.")
        layout = 0
    elif flow_layout == "nhwc":
        if tuple(flow.shape) != (n, h, w, 2):
            raise ValueError(f"The spatial sizes of input ({(h, w)}) and flow ({tuple(flow.shape)}) are not the same.")
        # a permuted view of an NCHW tensor (the reference passes flow.permute(0,2,3,1)) is used in place
        if not flow.is_contiguous() and flow.permute(0, 3, 1, 2).is_contiguous():
            flow = flow.permute(0, 3, 1, 2)
            layout = 0
        else:
            layout = 1
    else:
        raise ValueError(flow_layout)
    pm = {"zeros": 0, "border": 1, "reflection": 2}.get(padding_mode)
    if pm is None:
        raise ValueError(f"padding_mode={padding_mode!r}: 'zeros', 'border' or 'reflection'")
    if interpolation not in ("bilinear", "nearest"):
        raise ValueError(f"interpolation={interpolation!r}: 'bilinear' or 'nearest'")
    pm |= (0x10 if interpolation == "nearest" else 0) | (0 if align_corners else 0x20)
    flow = _chk(flow, "flow")
    if flow2 is not None:
        flow2 = _chk(flow2, "flow2")
        if flow2.shape != flow.shape:
            raise ValueError("flow2 must have the shape/layout of flow")
    out = torch.empty_like(x)
    st = _stream(x)
    _launch("flow_warp", 8.0 * n * c * h * w, 4.0 * n * h * w * (2 * c + 2 + (2 if flow2 is not None else 0)), x,
            lambda: lib().eavsr_flow_warp_f32(_p(x), _p(flow), _p(flow2), _p(out), n, c, h, w, layout, pm, st),
            "flow_warp")
    return out


def flow_warp_pair(xa: Tensor, xb: Tensor, flow: Tensor, flow2: Optional[Tensor] = None, b_il8=False):
    """(flow_warp(xa, flow [+ flow2]), flow_warp(xb, flow [+ flow2])) in one launch (networks.py:621,623); with b_il8 the
    second result comes in the IL8 layout (n, c/8, h, w, 8) of `dcnv2_il` -- fp32 (True) or rounded to 'fp16' / 'bf16' for
    `dcnv2_il16`."""
    xa, xb, flow = _chk(xa, "xa"), _chk(xb, "xb"), _chk(flow, "flow")
    n, c, h, w = xa.shape
    if xb.shape != xa.shape or tuple(flow.shape) != (n, 2, h, w):
        raise ValueError("flow_warp_pair: xa, xb (n,c,h,w) and flow (n,2,h,w)")
    if flow2 is not None:
        flow2 = _chk(flow2, "flow2")
        if flow2.shape != flow.shape:
            raise ValueError("flow2 must have the shape of flow")
    if b_il8 and c % 8:
        raise ValueError("flow_warp_pair: IL8 output needs c % 8 == 0")
    outa = torch.empty_like(xa)
    if isinstance(b_il8, str):
        code = h16_code(b_il8)                      # 1 fp16, 2 bf16
        outb = torch.empty((n, c // 8, h, w, 8), device=xa.device, dtype=_H16_TORCH[code])
        mode = 1 + code
    else:
        outb = torch.empty((n, c // 8, h, w, 8) if b_il8 else tuple(xb.shape), device=xa.device, dtype=torch.float32)
        mode = 1 if b_il8 else 0
    st = _stream(xa)
    _launch("flow_warp_pair", 16.0 * n * c * h * w, 4.0 * n * h * w * (4 * c + 2 + (2 if flow2 is not None else 0)), xa,
            lambda: lib().eavsr_flow_warp_pair_f32(_p(xa), _p(xb), _p(flow), _p(flow2), _p(outa), _p(outb), n, c, h, w,
                                                   mode, st), "flow_warp_pair")
    return outa, outb


# ------------------------------------------------------------------------------------------
# packed conv weights (cached per parameter version)
# ------------------------------------------------------------------------------------------
class _PackCache:
    """Packed weights, cached per weight *object* and version.  Entries hold weak references to the
    source tensors and are verified by identity, so a freed tensor whose id / address is reused
    can never produce a stale hit; nn.Parameters live as long as their module, so the hot path
    always hits."""

    def __init__(self):
        self._d = {}

    def get(self, weights: Sequence[Tensor]) -> Tensor:
        key = tuple((id(w), w._version) for w in weights)
        hit = self._d.get(key)
        if hit is not None:
            refs, packed = hit
            if all(r() is w for r, w in zip(refs, weights)):
                return packed
        w = weights[0] if len(weights) == 1 else torch.cat([x.detach() for x in weights], 0)
        w = _chk(w.detach(), "weight")
        cout, cin, kh, kw = w.shape
        if kh != kw:
            raise NotImplementedError("square kernels only")
        elems = lib().eavsr_packed_weight_elems(cout, cin, kh)
        if elems <= 0:
            raise NotImplementedError(f"conv weight shape {tuple(w.shape)} unsupported")
        packed = torch.empty(elems, device=w.device, dtype=torch.float32)
        with _DeviceOf(w):
            N.check(lib().eavsr_pack_conv_weight_f32(_p(w), _p(packed), cout, cin, kh, _stream(w)), "pack_conv_weight")
        d = self._d
        # drop stale versions of the same objects and dead entries
        ids = {id(x) for x in weights}
        for k in [k for k in d if any(i in ids for i, _ in k)]:
            d.pop(k, None)
        refs = tuple(weakref.ref(x, lambda _r, k=key, d=d: d.pop(k, None)) for x in weights)
        d[key] = (refs, packed)
        return packed

    def clear(self):
        self._d.clear()


pack_cache = _PackCache()

# Every cache of a form DERIVED from a parameter (packed / transformed / split / transposed weights, concatenated biases), keyed
# by (id, _version): graph.clear_weight_caches() empties all of them around a capture (a replayed graph updates the parameters
# without bumping `_version`).  A new cache registers itself here -- ADVICE r5: two caches added in round 5 were missing from
# the hand-kept list there.
WEIGHT_CACHES = [pack_cache]


def register_weight_cache(d):
    WEIGHT_CACHES.append(d)
    return d


def _cat_bias(biases: Sequence[Optional[Tensor]]) -> Optional[Tensor]:
    if biases[0] is None:
        return None
    return biases[0] if len(biases) == 1 else torch.cat([b.detach() for b in biases], 0)


_bias_cache = register_weight_cache({})


def _bias_of(biases: Sequence[Optional[Tensor]]) -> Optional[Tensor]:
    if len(biases) == 1:
        return None if biases[0] is None else _chk(biases[0].detach(), "bias")
    key = tuple((id(b), b._version) for b in biases)
    hit = _bias_cache.get(key)
    if hit is not None and all(r() is b for r, b in zip(hit[0], biases)):
        return hit[1]
    cat = _chk(_cat_bias(biases), "bias")
    refs = tuple(weakref.ref(b, lambda _r, k=key, c=_bias_cache: c.pop(k, None)) for b in biases)
    _bias_cache[key] = (refs, cat)
    return cat


_wcat_cache = register_weight_cache({})


def _cat_weights(weights: Sequence[Tensor]) -> Tensor:
    """cat(weights, 0) in the original layout, cached per (object, version) like the packed weights"""
    if len(weights) == 1:
        return _chk(weights[0].detach(), "weight")
    key = tuple((id(w), w._version) for w in weights)
    hit = _wcat_cache.get(key)
    if hit is not None and all(r() is w for r, w in zip(hit[0], weights)):
        return hit[1]
    cat = _chk(torch.cat([w.detach() for w in weights], 0), "weight")
    ids = {id(w) for w in weights}
    for k_ in [k_ for k_ in _wcat_cache if any(i in ids for i, _ in k_)]:
        _wcat_cache.pop(k_, None)
    refs = tuple(weakref.ref(w, lambda _r, k_=key, c=_wcat_cache: c.pop(k_, None)) for w in weights)
    _wcat_cache[key] = (refs, cat)
    return cat


# Small-cout 3x3 convolutions (64 -> 6 / 18 -> 2 heads of the pyramid levels, conv_last): eavsr_conv3x3_smallco_lite_f32 wherever the
# shape allows -- 30-39 registers, weights as scalar operands, so that its waves fit beside a resident Winograd workgroup of the other
# stream (48 registers / 28 KB are left on such a CU): conv_last 1,723 -> 850 us, and with 4 x 32-pixel tiles on the small launches
# the step 240.0 -> 236.8 ms.  EAVSR_SMALLCO=classic: round 2's 112-184-register kernels (the A/B reference).
SMALLCO_LITE = os.environ.get("EAVSR_SMALLCO", "lite") != "classic"
SMALLCO_LITE_MIN_TILES = 0
_smallco_pack_cache = register_weight_cache({})


def _packed_smallco(weights: Sequence[Tensor]) -> Tensor:
    """[ci][kx][block] form of a small-cout 3x3 weight (eavsr_pack_smallco_weight); cached per weight objects and versions"""
    key = tuple((id(w), w._version) for w in weights)
    hit = _smallco_pack_cache.get(key)
    if hit is not None and all(r() is w for r, w in zip(hit[0], weights)):
        return hit[1]
    w = _chk(_cat_weights(weights).detach(), "weight")
    cout, cin = int(w.shape[0]), int(w.shape[1])
    packed = torch.empty(lib().eavsr_smallco_packed_elems(cout, cin), device=w.device, dtype=torch.float32)
    with _DeviceOf(w):
        N.check(lib().eavsr_pack_smallco_weight(_p(w), _p(packed), cout, cin, _stream(w)), "pack_smallco_weight")
    ids = {id(x) for x in weights}
    for k in [k for k in _smallco_pack_cache if any(i in ids for i, _ in k)]:
        _smallco_pack_cache.pop(k, None)
    refs = tuple(weakref.ref(x, lambda _r, k=key, c=_smallco_pack_cache: c.pop(k, None)) for x in weights)
    _smallco_pack_cache[key] = (refs, packed)
    return packed


# 7x7 convolutions (SPyNet's basic module) and the 5x5 heads of the predictor: "bf16x6" = eavsr_conv_f32x6 (fp32 operands split
# exactly into bf16 terms, the contraction on the bf16 matrix pipe); otherwise the fp32-MFMA kernels (7x7: the implicit GEMM of
# eavsr_conv2d_f32; 5x5: F(2x2,5x5)).  A/B switches.
CONV7_MODE = os.environ.get("EAVSR_CONV7", "bf16x6")
CONV5_MODE = os.environ.get("EAVSR_CONV5", "bf16x6")
_conv7_pack_cache = register_weight_cache({})


def _packed_conv_x6(weights: Sequence[Tensor], dgrad: bool = False) -> Tensor:
    """A-operand form of a 7x7 / 5x5 / 3x3 weight (eavsr_pack_conv_weight_x6); cached per weight objects and versions.
    dgrad=True: the form of the INPUT-GRADIENT convolution of the (single) forward weight -- transposed and flipped by the pack
    kernel itself (eavsr_pack_conv_weight_x6_dgrad), no materialised copy."""
    key = tuple((id(w), w._version) for w in weights) + ((("dgrad", 0),) if dgrad else ())
    hit = _conv7_pack_cache.get(key)
    if hit is not None and all(r() is w for r, w in zip(hit[0], weights)):
        return hit[1]
    w = _chk(_cat_weights(weights).detach(), "weight")
    cout, cin, k = int(w.shape[0]), int(w.shape[1]), int(w.shape[-1])
    if dgrad:
        packed = torch.empty(lib().eavsr_conv_weight_x6_bytes(k, cin, cout), device=w.device, dtype=torch.uint8)
        with _DeviceOf(w):
            N.check(lib().eavsr_pack_conv_weight_x6_dgrad(_p(w), _p(packed), k, cout, cin, _stream(w)), "pack_conv_weight_x6_dgrad")
    else:
        packed = torch.empty(lib().eavsr_conv_weight_x6_bytes(k, cout, cin), device=w.device, dtype=torch.uint8)
        with _DeviceOf(w):
            N.check(lib().eavsr_pack_conv_weight_x6(_p(w), _p(packed), k, cout, cin, _stream(w)), "pack_conv_weight_x6")
    ids = {id(x) for x in weights}
    for k_ in [k_ for k_ in _conv7_pack_cache if (k_[-1] == ("dgrad", 0)) == dgrad and any(i in ids for i, _ in k_ if isinstance(i, int))]:
        _conv7_pack_cache.pop(k_, None)
    refs = tuple(weakref.ref(x, lambda _r, k_=key, c=_conv7_pack_cache: c.pop(k_, None)) for x in weights)
    _conv7_pack_cache[key] = (refs, packed)
    return packed


def prepack_conv3_x6(weights: Sequence[Tensor]) -> int:
    """Both packed forms (forward and input-gradient) of every (64, 64, 3, 3) weight in `weights` that the cache does not hold at
    the weight's current version, in ceil(count / 48) launches (eavsr_pack_conv_weight_x6_multi) instead of one launch per weight and
    form at its first use: what the training step calls once, in front of its forward (540 launches of 4.7 us on its one dependent
    chain otherwise).  Returns the number of forms packed."""
    todo = []
    for w in weights:
        if tuple(w.shape) != (64, 64, 3, 3) or not w.is_cuda or not w.is_contiguous() or w.dtype != torch.float32:
            continue
        for dg in (False, True):
            key = ((id(w), w._version),) + ((("dgrad", 0),) if dg else ())
            hit = _conv7_pack_cache.get(key)
            if hit is None or hit[0][0]() is not w:
                todo.append((w, dg, key))
    if not todo:
        return 0
    dev = todo[0][0].device
    nbytes = int(lib().eavsr_conv_weight_x6_bytes(3, 64, 64))
    store = torch.empty((len(todo), nbytes), device=dev, dtype=torch.uint8)
    cnt = len(todo)
    srcs = (C.c_void_p * cnt)(*[w.detach().data_ptr() for w, _, _ in todo])
    dsts = (C.c_void_p * cnt)(*[store[i].data_ptr() for i in range(cnt)])
    trs = (C.c_int32 * cnt)(*[int(dg) for _, dg, _ in todo])
    with _DeviceOf(todo[0][0]):
        N.check(lib().eavsr_pack_conv_weight_x6_multi(srcs, dsts, trs, cnt, 3, 64, _stream(todo[0][0])), "pack_conv_weight_x6_multi")
    for i, (w, dg, key) in enumerate(todo):
        for k_ in [k_ for k_ in _conv7_pack_cache if (k_[-1] == ("dgrad", 0)) == dg and any(i_ == id(w) for i_, _ in k_ if isinstance(i_, int))]:
            _conv7_pack_cache.pop(k_, None)
        refs = (weakref.ref(w, lambda _r, k_=key, c=_conv7_pack_cache: c.pop(k_, None)),)
        _conv7_pack_cache[key] = (refs, store[i])
    return cnt


def x6s_takes(n: int, h: int, w: int) -> bool:
    """a single-source 3x3 64 -> 64 launch of this size runs on eavsr_conv3x3_f32x6s (the crop-sized bf16x6 kernel) in the current mode"""
    return (CONV3_SMALL == "x6s" and CONV_MODE in ("winograd", "winograd4") and CONV3_H16 is None
            and n * lib().eavsr_conv3x3_x6s_tiles(h, w) <= X6S_MAX_TILES)


def _conv_x6(x: Tensor, weights, biases, act, slope, sigmoid_from: int = -1):
    n, cin, h, w = x.shape
    cout = sum(int(w_.shape[0]) for w_ in weights)
    k = int(weights[0].shape[-1])
    wp = _packed_conv_x6(weights)
    b = _bias_of(biases)
    out = torch.empty((n, cout, h, w), device=x.device, dtype=torch.float32)
    st = _stream(x)
    px = float(n * h * w)
    _launch(f"conv{k}x{k}_{cin}to{cout}_x6", 2.0 * cin * cout * k * k * px, 4.0 * px * (cin + cout), x,
            lambda: lib().eavsr_conv_f32x6(_p(x), _p(wp), _p(b), _p(out), n, cin, cout, h, w, k, ACT[act], float(slope),
                                           int(sigmoid_from), st),
            "conv_f32x6")
    return out


# The 16-bit modes' generic 3x3 convolution (csrc/conv3_h16.hip): None = off (fp32 everywhere, the default), "bf16" / "fp16" =
# every plain 3x3 convolution with >= 32 output channels whose sources have channel counts % 16 == 0 rounds its operands to that
# type (fp32 NCHW in and out).  Set by networks.set_backbone_dtype (EAVSR_CONV3_16BIT=0 keeps these convolutions fp32: A/B switch).
CONV3_H16 = None
CONV3_H16_ENABLED = os.environ.get("EAVSR_CONV3_16BIT", "1") == "1"
_h16g_pack_cache = register_weight_cache({})


def set_conv3_h16(dtype) -> None:
    global CONV3_H16
    if dtype is not None:
        h16_code(dtype)
    CONV3_H16 = dtype if CONV3_H16_ENABLED else None


def _packed_h16g(weights: Sequence[Tensor], code: int) -> Tensor:
    key = tuple((id(w), w._version) for w in weights) + (code,)
    hit = _h16g_pack_cache.get(key)
    if hit is not None and all(r() is w for r, w in zip(hit[0], weights)):
        return hit[1]
    w = _chk(_cat_weights(weights).detach(), "weight")
    cout, cin = int(w.shape[0]), int(w.shape[1])
    packed = torch.empty(int(lib().eavsr_conv3x3_h16g_weight_bytes(cout, cin)), device=w.device, dtype=torch.uint8)
    with _DeviceOf(w):
        N.check(lib().eavsr_pack_conv3x3_h16g(_p(w), _p(packed), cout, cin, code, _stream(w)), "pack_conv3x3_h16g")
    ids = {id(x) for x in weights}
    for k_ in [k_ for k_ in _h16g_pack_cache if any(isinstance(e, tuple) and e[0] in ids for e in k_)]:
        _h16g_pack_cache.pop(k_, None)
    refs = tuple(weakref.ref(x, lambda _r, k_=key, c=_h16g_pack_cache: c.pop(k_, None)) for x in weights)
    _h16g_pack_cache[key] = (refs, packed)
    return packed


# SPyNet's 7x7 layers with >= 16 output channels in the 16-bit modes (one operand plane of csrc/conv_x6.hip); EAVSR_CONV7_16BIT=0
# keeps them on the exact bf16x6 form (A/B switch)
CONV7_H16_ENABLED = os.environ.get("EAVSR_CONV7_16BIT", "1") == "1"
_h16x1_pack_cache = register_weight_cache({})


def _conv_h16x1(x: Tensor, weights, biases, act, slope, dtype):
    n, cin, h, w = x.shape
    cout = sum(int(w_.shape[0]) for w_ in weights)
    k = int(weights[0].shape[-1])
    code = h16_code(dtype)
    key = tuple((id(w_), w_._version) for w_ in weights) + (code,)
    hit = _h16x1_pack_cache.get(key)
    if hit is not None and all(r() is w_ for r, w_ in zip(hit[0], weights)):
        wp = hit[1]
    else:
        wc = _chk(_cat_weights(weights).detach(), "weight")
        wp = torch.empty(lib().eavsr_conv_weight_h16x1_bytes(k, cout, cin), device=wc.device, dtype=torch.uint8)
        with _DeviceOf(wc):
            N.check(lib().eavsr_pack_conv_weight_h16x1(_p(wc), _p(wp), k, cout, cin, code, _stream(wc)), "pack_conv_weight_h16x1")
        ids = {id(x_) for x_ in weights}
        for k_ in [k_ for k_ in _h16x1_pack_cache if any(isinstance(e, tuple) and e[0] in ids for e in k_)]:
            _h16x1_pack_cache.pop(k_, None)
        refs = tuple(weakref.ref(x_, lambda _r, k_=key, c=_h16x1_pack_cache: c.pop(k_, None)) for x_ in weights)
        _h16x1_pack_cache[key] = (refs, wp)
    b = _bias_of(biases)
    out = torch.empty((n, cout, h, w), device=x.device, dtype=torch.float32)
    st = _stream(x)
    px = float(n * h * w)
    _launch(f"conv{k}x{k}_{cin}to{cout}_h16x1", 2.0 * cin * cout * k * k * px, 4.0 * px * (cin + cout), x,
            lambda: lib().eavsr_conv_h16x1(_p(x), _p(wp), _p(b), _p(out), n, cin, cout, h, w, k, ACT[act], float(slope), code, st),
            "conv_h16x1")
    return out


def _conv3x3_h16g(srcs, weights, biases, act, slope, dtype):
    n, _, h, w = srcs[0].shape
    cin = sum(int(s_.shape[1]) for s_ in srcs)
    cout = sum(int(w_.shape[0]) for w_ in weights)
    code = h16_code(dtype)
    wp = _packed_h16g(weights, code)
    b = _bias_of(biases)
    out = torch.empty((n, cout, h, w), device=srcs[0].device, dtype=torch.float32)
    d = N.ConvDesc()
    for i in range(5):
        d.src[i] = _p(srcs[i]) if i < len(srcs) else None
        d.src_c[i] = int(srcs[i].shape[1]) if i < len(srcs) else 0
    d.n_src = len(srcs)
    d.ksize = 3
    d.weight_packed = _p(wp)
    d.bias = _p(b)
    d.out = _p(out)
    d.n, d.h, d.w, d.cin, d.cout = n, h, w, cin, cout
    d.act = ACT[act]
    d.slope = float(slope)
    st = _stream(out)
    px = float(n) * h * w
    _launch(f"conv3x3_{cin}to{cout}_h16g", 2.0 * cin * cout * 9 * px, 4.0 * px * (cin + cout), out,
            lambda: lib().eavsr_conv3x3_h16g_f32(C.byref(d), code, st), "conv3x3_h16g")
    return out


def _conv3x3_smallco(x: Tensor, weights, biases, act, slope, residual):
    n, cin, h, w = x.shape
    wt = _cat_weights(weights)
    cout = int(wt.shape[0])
    b = _bias_of(biases)
    out = torch.empty((n, cout, h, w), device=x.device, dtype=torch.float32)
    if residual is not None:
        residual = _chk(residual, "residual")
        if residual.shape != out.shape:
            raise ValueError("residual shape mismatch")
    st = _stream(x)
    px = float(n) * h * w
    if (SMALLCO_LITE and cout in (2, 3, 4, 6) and w % 4 == 0 and x.data_ptr() % 16 == 0 and h * w * cin * 4 < 2 ** 32
            and n * ((h + 7) // 8) * ((w + 63) // 64) >= SMALLCO_LITE_MIN_TILES):
        wp = _packed_smallco(list(weights))
        _launch(f"conv3x3_{cin}to{cout}", 2.0 * cin * cout * 9 * px, 4.0 * px * (cin + cout + (cout if residual is not None else 0)), x,
                lambda: lib().eavsr_conv3x3_smallco_lite_f32(_p(x), _p(wp), _p(b), _p(residual), _p(out), n, cin, h, w, cout,
                                                             ACT[act], float(slope), st), "conv3x3_smallco_lite")
        return out
    _launch(f"conv3x3_{cin}to{cout}", 2.0 * cin * cout * 9 * px, 4.0 * px * (cin + cout + (cout if residual is not None else 0)), x,
            lambda: lib().eavsr_conv3x3_smallco_f32(_p(x), _p(wt), _p(b), _p(residual), _p(out), n, cin, h, w, cout,
                                                    ACT[act], float(slope), st), "conv3x3_smallco")
    return out


# ------------------------------------------------------------------------------------------
# dense conv  (nn.Conv2d, stride 1, "same" padding)
# ------------------------------------------------------------------------------------------
def _wino_fusable(x: Tensor) -> bool:
    return (lab_available() and CONV_MODE in ("winograd", "winograd4") and x.dim() == 4 and x.shape[3] % 4 == 0 and x.shape[1] % 8 == 0 and x.shape[1] <= 256
            and x.is_contiguous() and x.data_ptr() % 16 == 0
            and int(x.shape[0]) * _wino_tiles(int(x.shape[2]), int(x.shape[3])) >= WINO_MIN_TILES)


def ca_fusable(x: Tensor, cout: int = 64) -> bool:
    """Can the channel-attention tail `r * scale + x` be folded into the next 3x3 conv?  Yes where the Winograd kernel
    runs (its input transform applies it), or on the direct kernel's 32-row, 16-byte-DMA path."""
    if _wino_fusable(x):
        return True
    return (x.dim() == 4 and x.shape[3] % 4 == 0 and x.shape[1] % 4 == 0 and x.shape[1] <= 256 and 32 < cout <= 64
            and x.is_contiguous() and x.data_ptr() % 16 == 0
            and lib().eavsr_conv2d_tile_rows(int(x.shape[0]), int(x.shape[2]), int(x.shape[3]), 3) == 32)


# EAVSR_FUSE_SHUFFLE=0: the upsampling tail's PixelShuffle(2) as a torch copy instead of the conv kernel's store pattern (A/B switch)
FUSE_PIXEL_SHUFFLE = os.environ.get("EAVSR_FUSE_SHUFFLE", "1") == "1"


_dgrad_w_cache = register_weight_cache({})


def dgrad_weight(w: Tensor) -> Tensor:
    """(cin, cout, k, k) transposed and flipped copy of a forward weight: the weight of its input-gradient convolution; cached
    per weight object and version"""
    key = (id(w), w._version)
    hit = _dgrad_w_cache.get(key)
    if hit is not None and hit[0]() is w:
        return hit[1]
    wt = w.detach().flip(2, 3).transpose(0, 1).contiguous()
    for k_ in [k_ for k_ in _dgrad_w_cache if k_[0] == id(w)]:
        _dgrad_w_cache.pop(k_, None)
    _dgrad_w_cache[key] = (weakref.ref(w, lambda _r, k_=key, c=_dgrad_w_cache: c.pop(k_, None)), wt)
    return wt


class BorderPieces:
    """what conv2d(.., border=True) leaves beside the per-tile channel sums: `data` (n, 4, stride, 64) -- border 0 / 1 = image row 0 /
    h - 1 (`p_rows` pieces), 2 / 3 = image column 0 / w - 1 (`p_cols` pieces) of the convolution's OUTPUT (desc.border_pieces)"""
    __slots__ = ("data", "p_rows", "p_cols")

    def __init__(self, data: Tensor, p_rows: int, p_cols: int):
        self.data, self.p_rows, self.p_cols = data, int(p_rows), int(p_cols)


def conv2d(srcs, weight, bias=None, act=None, slope: float = 0.0, residual=None, chan_partial: bool = False, ca=None, ca_out: bool = False,
           pixel_shuffle2: bool = False, sigmoid_from=None, dgrad: bool = False, res_scale=None, border: bool = False, sum_mul=None):
    """see _conv2d (the implementation); this shim only normalises the `border` result: routes that do not produce border pieces
    return None in its place"""
    r = _conv2d(srcs, weight, bias, act, slope, residual, chan_partial, ca, ca_out, pixel_shuffle2, sigmoid_from, dgrad, res_scale, border,
                sum_mul)
    if border and not (isinstance(r, tuple) and len(r) >= 2 and (r[-1] is None or isinstance(r[-1], BorderPieces))):
        r = (tuple(r) if isinstance(r, tuple) else (r,)) + (None,)
    return r


def _conv2d(srcs: Union[Tensor, Sequence[Tensor]], weight: Union[Tensor, Sequence[Tensor]],
           bias: Union[None, Tensor, Sequence[Optional[Tensor]]] = None, act: Optional[str] = None,
           slope: float = 0.0, residual: Optional[Tensor] = None, chan_partial: bool = False,
           ca: Optional[Tuple[Tensor, Tensor]] = None, ca_out: bool = False, pixel_shuffle2: bool = False,
           sigmoid_from: Optional[int] = None, dgrad: bool = False, res_scale: Optional[Tensor] = None, border: bool = False,
           sum_mul: Optional[Tensor] = None):
    """conv over the virtual channel-concatenation of `srcs`; `weight` may be a list of weights
    that are concatenated along cout (several heads in one launch).
    border=True (with chan_partial=True): a third result, the sums of the output's four border lines per border tile
    (`BorderPieces`, for ca_scale_pre(.., border=)) where the grouped F(4x4,3x3) kernel runs, None on every other route.
    res_scale (n, cout) with `residual`: out = residual + res_scale[n, co] * act(conv + bias) -- RCABlock's tail as the epilogue of
    the F(4x4,3x3) kernel (the attention from ca_scale_pre BEFORE the launch); other routes: the convolution, then scale_residual.
    dgrad=True: `weight` is ONE forward weight (cout_w, cin_w, k, k) and the call computes the input gradient of its stride-1
    "same" convolution from srcs = dY (cout_w channels): the convolution with the transposed, flipped weight.  The small-launch
    bf16x6 kernel packs that form straight from `weight`; every other route materialises it once per weight version.
    sigmoid_from=c: output channels >= c (a multiple of 8) leave through the sigmoid instead of `act` (the mask head of the
    predictor, networks.py:313-314) -- in the epilogue of the bf16x6 5x5 / 7x7 kernel, by torch on every other route.
    pixel_shuffle2=True returns F.pixel_shuffle(out, 2) -- written by the F(4x4,3x3) kernel's epilogue itself where that kernel
    runs (the upsampling tail, eavsrp_model.py:343-347), by torch otherwise.
    ca=(scale (n,c), x (n,c,h,w)): the conv input is srcs * scale[n,c] + x (RCABlock tail fused into this
    conv); with ca_out=True that effective input is also returned.
    sum_mul=m (n, cout, h, w), with dgrad: also returns (n, rows, cout) partial sums over the plane of out * m -- `sum_hw d r`, what the
    backward of the PREVIOUS RCAB's tail starts with -- from the epilogue of the small-launch bf16x6 kernel (desc.sum_mul), by a
    plane-sum launch behind every other kernel (rows = 1).
    Returns out, then the per-tile channel sums when chan_partial=True, then the effective input when
    ca_out=True."""
    if sum_mul is not None:
        if not dgrad or chan_partial or ca is not None or pixel_shuffle2 or sigmoid_from is not None or res_scale is not None or act == "relu_mask":
            raise ValueError("sum_mul: input-gradient convolutions only (dgrad=True; no channel sums / prologue / shuffle / sigmoid / mask)")
        sum_mul = _chk(sum_mul, "sum_mul")
    if isinstance(srcs, torch.Tensor):
        srcs = [srcs]
    weights = [weight] if isinstance(weight, torch.Tensor) else list(weight)
    biases = [bias] if (bias is None or isinstance(bias, torch.Tensor)) else list(bias)
    srcs = [_chk(s, f"src{i}") for i, s in enumerate(srcs)]
    if not 1 <= len(srcs) <= 5:
        raise ValueError("1..5 sources")
    n, _, h, w = srcs[0].shape
    for s in srcs:
        if s.shape[0] != n or tuple(s.shape[2:]) != (h, w):
            raise ValueError("all sources must share n,h,w")
    cin = sum(int(s.shape[1]) for s in srcs)
    if dgrad:
        if len(weights) != 1 or biases != [None] or chan_partial or ca is not None or pixel_shuffle2 or sigmoid_from is not None:
            raise ValueError("dgrad: one forward weight, no bias / channel sums / prologue / shuffle / sigmoid")
        if int(weights[0].shape[0]) != cin:
            raise ValueError(f"dgrad: the forward weight has {int(weights[0].shape[0])} output channels, dY gives {cin}")
        k_ = int(weights[0].shape[-1])
        # only the small-launch bf16x6 kernel reads the forward weight in place (necessary conditions; the decision is below)
        if not (CONV3_SMALL == "x6s" and CONV_MODE in ("winograd", "winograd4") and k_ == 3 and len(srcs) == 1 and cin == 64
                and int(weights[0].shape[1]) not in (2, 3, 4, 6) and CONV3_H16 is None
                and n * lib().eavsr_conv3x3_x6s_tiles(h, w) <= X6S_MAX_TILES):
            y = conv2d(srcs, dgrad_weight(weights[0]), None, act, slope, residual)
            return y if sum_mul is None else (y, plane_sum(y, sum_mul).view(n, 1, -1))
    cout = sum(int(x.shape[0]) for x in weights) if not dgrad else int(weights[0].shape[1])
    k = int(weights[0].shape[-1])
    if not dgrad and any(int(x.shape[1]) != cin for x in weights):
        raise ValueError(f"weight expects {[int(x.shape[1]) for x in weights]} input channels, sources give {cin}")
    if pixel_shuffle2 and (cout % 4 or residual is not None or chan_partial or ca is not None):
        raise ValueError("pixel_shuffle2: cout % 4 == 0, no residual / channel sums / channel-attention prologue")
    if res_scale is not None:
        if residual is None or chan_partial or ca is not None or pixel_shuffle2 or sigmoid_from is not None or dgrad or act == "relu_mask":
            raise ValueError("res_scale: with `residual`; no channel sums / prologue / shuffle / sigmoid / dgrad / mask")
        res_scale = _chk(res_scale, "res_scale")
        if tuple(res_scale.shape) != (n, cout):
            raise ValueError(f"res_scale must be (n, cout) = ({n}, {cout})")
    masked = act == "relu_mask"    # out = residual > 0 ? conv : 0 (a ReLU's backward inside the input-gradient convolution): direct kernels only
    if masked and (residual is None or chan_partial or ca is not None or pixel_shuffle2 or sigmoid_from is not None):
        raise ValueError("act='relu_mask': residual = the ReLU's forward output; no channel sums / prologue / shuffle / sigmoid")
    if ca_out and ca is None:      # before any early return (the bf16x6 route used to skip this check)
        raise ValueError("ca_out needs ca")
    if sigmoid_from is not None:   # validated once, for every route (routes used to differ)
        sigmoid_from = int(sigmoid_from)
        if not 0 <= sigmoid_from < cout:
            raise ValueError(f"sigmoid_from {sigmoid_from}: 0 <= value < cout = {cout}")
        if residual is not None or chan_partial or ca is not None or pixel_shuffle2:
            raise ValueError("sigmoid_from: plain convolutions only")
        fused = (((k == 7 and CONV7_MODE == "bf16x6") or (k == 5 and CONV5_MODE == "bf16x6")) and len(srcs) == 1
                 and cin % 8 == 0 and sigmoid_from % 8 == 0)
        if not fused:              # every other kernel: the activation is one more pass over those channels
            y = conv2d(srcs, weights, biases, act, slope)
            y[:, sigmoid_from:] = torch.sigmoid(y[:, sigmoid_from:])
            return y
    if k == 3 and len(srcs) == 1 and cout in (2, 3, 4, 6) and not chan_partial and ca is None and not masked and res_scale is None:
        y = _conv3x3_smallco(srcs[0], weights, biases, act, slope, residual)
        return torch.nn.functional.pixel_shuffle(y, 2) if pixel_shuffle2 else y
    if (CONV3_H16 is not None and k == 7 and CONV7_H16_ENABLED and sigmoid_from is None and len(srcs) == 1 and cin % 8 == 0 and cout >= 16
            and residual is None and not chan_partial and ca is None and not pixel_shuffle2 and not torch.is_grad_enabled()):
        return _conv_h16x1(srcs[0], weights, biases, act, slope, CONV3_H16)      # SPyNet's feature layers (its 16 -> 2 flow head stays exact)
    if (((k == 7 and CONV7_MODE == "bf16x6") or (k == 5 and CONV5_MODE == "bf16x6")) and len(srcs) == 1 and cin % 8 == 0
            and residual is None and not chan_partial and ca is None and not pixel_shuffle2):
        return _conv_x6(srcs[0], weights, biases, act, slope, -1 if sigmoid_from is None else int(sigmoid_from))
    if (CONV3_H16 is not None and k == 3 and sigmoid_from is None and residual is None and not chan_partial and ca is None
            and not pixel_shuffle2 and w % 4 == 0 and cout >= 32 and not torch.is_grad_enabled()
            and all(int(s_.shape[1]) % 16 == 0 and s_.data_ptr() % 16 == 0 for s_ in srcs)):
        return _conv3x3_h16g(srcs, weights, biases, act, slope, CONV3_H16)
    ck = lib().eavsr_conv2d_ck(k)
    if any(int(s.shape[1]) % ck for s in srcs[:-1]):
        srcs = [torch.cat(srcs, 1)]  # ragged middle source: materialise (tiny SPyNet inputs only)
    b = _bias_of(biases)
    out = torch.empty((n, cout, h, w), device=srcs[0].device, dtype=torch.float32)
    base_ok = (k == 3 and w % 4 == 0 and all(int(s_.shape[1]) % 8 == 0 and s_.data_ptr() % 16 == 0 for s_ in srcs))
    x9_ok = base_ok and ca is None and not masked
    if CONV_MODE in ("winograd", "bf16x9"):      # (set through the environment: set_conv_mode() checks at the call)
        require_lab(f"conv mode {CONV_MODE!r}")
    lab = lab_available()
    use_wino = (CONV_MODE in ("winograd", "winograd4") and base_ok and not masked
                and n * _wino_tiles(h, w) >= WINO_MIN_TILES
                and (ca is None or (lab and len(srcs) == 1 and cin <= 256 and ca[1].data_ptr() % 16 == 0)))
    # F(4x4, 3x3): same 512-pixel-per-workgroup granularity (8 x 64), no fused channel-attention prologue
    use_wino4 = (use_wino and CONV_MODE == "winograd4" and out.data_ptr() % 16 == 0
                 and (residual is None or residual.data_ptr() % 16 == 0)
                 and 2 * n * lib().eavsr_conv3x3_wino4_tiles(h, w) >= WINO_MIN_TILES)
    if use_wino and not use_wino4 and not lab:
        use_wino = False      # default build: F(2x2,3x3) is a lab kernel -- what F(4x4,3x3) does not take runs the direct kernel
    if res_scale is not None and not (use_wino4 and cin % 8 == 0 and lib().eavsr_wino4_schedule() >= 1):
        # every other kernel: the convolution, then the tail as its own launch
        return scale_residual(conv2d(srcs, weights, biases, act=act, slope=slope), res_scale, residual)
    if pixel_shuffle2 and not (use_wino4 and FUSE_PIXEL_SHUFFLE):      # every other kernel: plain output, shuffled by torch
        return torch.nn.functional.pixel_shuffle(conv2d(srcs, weights, biases, act=act, slope=slope), 2)
    # 5x5 (the predictor's offset / mask heads) by F(2x2, 5x5): the same 6 x 6 tile pipeline, 4 x 32-pixel tiles
    use_wino5 = (CONV_MODE == "winograd4" and k == 5 and ca is None and not masked and w % 4 == 0 and out.data_ptr() % 8 == 0
                 and all(int(s_.shape[1]) % 4 == 0 and s_.data_ptr() % 16 == 0 for s_ in srcs)
                 and (residual is None or residual.data_ptr() % 8 == 0)
                 and n * lib().eavsr_conv5x5_wino_tiles(h, w) * ((cout + 63) // 64) >= 2 * WINO_MIN_TILES)
    # small launches of the 64-channel 3x3 convolution (a training crop, a pyramid level): exact bf16x6 instead of the fp32 MFMA
    # (the explicit modes "direct" / "bf16x9" keep their own kernels: they are the A/B references)
    use_x6s = (CONV3_SMALL == "x6s" and CONV_MODE in ("winograd", "winograd4") and k == 3 and len(srcs) == 1 and cin == 64
               and ca is None and not pixel_shuffle2 and not use_wino
               and n * lib().eavsr_conv3x3_x6s_tiles(h, w) <= X6S_MAX_TILES)
    if sum_mul is not None and not (use_x6s and w % 4 == 0 and tuple(sum_mul.shape) == (n, cout, h, w) and
                                    all(t_ is None or t_.data_ptr() % 16 == 0 for t_ in (srcs[0], out, residual, sum_mul))):
        y = conv2d(srcs, weights, None, act, slope, residual, dgrad=True)      # the kernel's epilogue form does not apply: a launch
        return y, plane_sum(y, sum_mul).view(n, 1, -1)
    if dgrad and not use_x6s:      # (e.g. a large launch that the Winograd kernel takes)
        weights, dgrad = [dgrad_weight(weights[0])], False
    wp = None if use_x6s else pack_cache.get(weights)      # (the x6s kernel has its own packed form)
    part = None
    if sum_mul is not None:      # (use_x6s holds)
        part = torch.empty((n, lib().eavsr_conv3x3_x6s_tiles(h, w), cout), device=out.device, dtype=torch.float32)
    if chan_partial:
        tiles = (lib().eavsr_conv3x3_x6s_tiles(h, w) if use_x6s else lib().eavsr_conv5x5_wino_tiles(h, w) if use_wino5 else lib().eavsr_conv3x3_wino4_tiles(h, w) if use_wino4 else _wino_tiles(h, w) if use_wino
                 else lib().eavsr_conv2d_tiles(n, h, w, k))
        part = torch.empty((n, tiles, cout), device=out.device, dtype=torch.float32)
    if residual is not None:
        residual = _chk(residual, "residual")
        if residual.shape != out.shape:
            raise ValueError("residual shape mismatch")
    d = N.ConvDesc()
    for i in range(5):
        d.src[i] = _p(srcs[i]) if i < len(srcs) else None
        d.src_c[i] = int(srcs[i].shape[1]) if i < len(srcs) else 0
    d.n_src = len(srcs)
    d.ksize = k
    d.weight_packed = _p(wp) if wp is not None else None
    d.bias = _p(b)
    d.residual = _p(residual)
    d.out = _p(out)
    d.chan_partial = _p(part)
    d.n, d.h, d.w, d.cin, d.cout = n, h, w, cin, cout
    d.act = ACT[act]
    d.slope = float(slope)
    d.res_scale = _p(res_scale)
    d.sum_mul = _p(sum_mul)
    pieces = None
    if (border and chan_partial and use_wino4 and cout == 64 and cin % 8 == 0 and ca is None and not pixel_shuffle2 and res_scale is None
            and lib().eavsr_wino4_schedule() == 1):
        pr, pc = C.c_int32(0), C.c_int32(0)
        N.check(lib().eavsr_conv3x3_wino4_border_pieces(h, w, C.byref(pr), C.byref(pc)), "conv3x3_wino4_border_pieces")
        stride = max(pr.value, pc.value)
        pieces = BorderPieces(torch.empty((n, 4, stride, 64), device=out.device, dtype=torch.float32), pr.value, pc.value)
        d.border_pieces, d.border_stride = _p(pieces.data), stride
    if pixel_shuffle2 and use_wino4:
        d.out_shuffle = 2
        out = out.view(n, cout // 4, 2 * h, 2 * w)      # the same buffer, written in the shuffled layout
    xs = None
    if ca is not None:
        scale, cx = _chk(ca[0], "ca scale"), _chk(ca[1], "ca x")
        if len(srcs) != 1 or cx.shape != srcs[0].shape or tuple(scale.shape) != (n, cin):
            raise ValueError("ca=(scale, x): one source, x of the same shape, scale (n, cin)")
        if not use_wino and (k != 3 or not ca_fusable(srcs[0], cout) or cx.data_ptr() % 16):
            raise NotImplementedError("fused channel-attention prologue unavailable for this shape; use "
                                      "scale_residual + conv2d (see ops.ca_fusable)")
        d.ca_scale, d.ca_x = _p(scale), _p(cx)
        if ca_out:
            xs = torch.empty_like(cx)
            d.ca_out = _p(xs)
    elif ca_out:
        raise ValueError("ca_out needs ca")
    st = _stream(out)
    px = float(n) * h * w
    if use_x6s:
        wq = _packed_conv_x6(weights, dgrad=dgrad)
        _launch(f"conv3x3_{cin}to{cout}_x6s", 2.0 * cin * cout * 9 * px,
                4.0 * px * (cin + cout + (cout if residual is not None else 0)), out,
                lambda: lib().eavsr_conv3x3_f32x6s(C.byref(d), _p(wq), st), "conv3x3_f32x6s")
        return out if not (chan_partial or sum_mul is not None) else (out, part)
    if use_wino5:
        wu = _packed_wino(weights, kind="f5")
        _launch(f"conv5x5_{cin}to{cout}_wino", 2.0 * cin * cout * 25 * px,
                4.0 * px * (cin + cout + (cout if residual is not None else 0)), out,
                lambda: lib().eavsr_conv5x5_wino_f32(C.byref(d), _p(wu), st), "conv5x5_wino")
        return out if not chan_partial else (out, part)
    if use_wino4:
        wu = _packed_wino(weights, four=True)
        _launch(f"conv3x3_{cin}to{cout}_wino4" + ("_ca" if ca is not None else ""), 2.0 * cin * cout * 9 * px,
                4.0 * px * (cin * (1 if ca is None else (3 if ca_out else 2)) + cout + (cout if residual is not None else 0)), out,
                lambda: lib().eavsr_conv3x3_wino4_f32(C.byref(d), _p(wu), st), "conv3x3_wino4")
        res = [out] + ([part] if chan_partial else []) + ([xs] if xs is not None else []) + ([pieces] if border else [])
        return res[0] if len(res) == 1 else tuple(res)
    if use_wino:
        wu = _packed_wino(weights)
        _launch(f"conv3x3_{cin}to{cout}_wino" + ("_ca" if ca is not None else ""), 2.0 * cin * cout * 9 * px,
                4.0 * px * (cin * (1 if ca is None else (3 if ca_out else 2)) + cout + (cout if residual is not None else 0)), out,
                lambda: lib().eavsr_conv3x3_wino_f32(C.byref(d), _p(wu), st), "conv3x3_wino")
        res = [out] + ([part] if chan_partial else []) + ([xs] if xs is not None else [])
        return res[0] if len(res) == 1 else tuple(res)
    if CONV_MODE == "bf16x9" and x9_ok and lib().eavsr_conv2d_tile_rows(n, h, w, 3) == 32:
        wx = _packed_x9(weights)
        _launch(f"conv3x3_{cin}to{cout}_x9", 2.0 * cin * cout * 9 * px,
                4.0 * px * (cin + cout + (cout if residual is not None else 0)), out,
                lambda: lib().eavsr_conv3x3_f32x9(C.byref(d), _p(wx), st), "conv3x3_f32x9")
        return out if not chan_partial else (out, part)
    _launch(f"conv{k}x{k}_{cin}to{cout}" + ("_ca" if ca is not None else ""), 2.0 * cin * cout * k * k * px,
            4.0 * px * (cin * (1 if ca is None else (3 if ca_out else 2)) + cout + (cout if residual is not None else 0)), out,
            lambda: lib().eavsr_conv2d_f32(C.byref(d), st), "conv2d")
    if not chan_partial and xs is None:
        return out
    res = [out]
    if chan_partial:
        res.append(part)
    if xs is not None:
        res.append(xs)
    return tuple(res)


# ------------------------------------------------------------------------------------------
# DCNv2  (mmcv.ops.modulated_deform_conv2d as called at networks.py:627-630)
# ------------------------------------------------------------------------------------------
def _one(v):
    return int(v[0]) if isinstance(v, (tuple, list)) else int(v)


def modulated_deform_conv2d(input: Tensor, offset: Tensor, mask: Tensor, weight: Tensor,
                            bias: Optional[Tensor] = None, stride=1, padding=0, dilation=1, groups=1,
                            deform_groups=1) -> Tensor:
    """Same signature as mmcv.ops.modulated_deform_conv2d.  The reference's configuration (3x3, stride 1, padding 1,
    dilation 1, groups 1, channels per deformable group a multiple of 8: networks.py:577-583 + eavsrp_model.py:143) runs
    the fused LDS-window MFMA kernel; every other configuration of the signature runs eavsr_dcnv2_generic_f32 (a plain
    kernel, forward only).  There is no CPU path."""
    x = _chk(input, "input")
    cout, cin_g, kh, kw = (int(v) for v in weight.shape)
    n, cin, h, w = x.shape
    two = lambda v: (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))
    (sh, sw), (ph, pw), (dh, dw) = two(stride), two(padding), two(dilation)
    if cin_g * groups != cin:
        raise ValueError("weight / input channel mismatch")
    if cin % deform_groups or cout % groups:
        raise ValueError("channels not divisible by groups / deform_groups")
    ho = (h + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    wo = (w + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    offset = _chk(offset, "offset")
    mask = _chk(mask, "mask")
    if tuple(offset.shape) != (n, deform_groups * 2 * kh * kw, ho, wo):
        raise ValueError(f"offset shape {tuple(offset.shape)} != {(n, deform_groups * 2 * kh * kw, ho, wo)}")
    if tuple(mask.shape) != (n, deform_groups * kh * kw, ho, wo):
        raise ValueError(f"mask shape {tuple(mask.shape)} != {(n, deform_groups * kh * kw, ho, wo)}")
    b = None if bias is None else _chk(bias.detach(), "bias")
    out = torch.empty((n, cout, ho, wo), device=x.device, dtype=torch.float32)
    on_path = ((kh, kw) == (3, 3) and (sh, sw, ph, pw, dh, dw) == (1, 1, 1, 1, 1, 1) and groups == 1
               and (cin // deform_groups) % 8 == 0)
    if not on_path:
        if groups > 1 and (cout // groups) % 8:
            raise NotImplementedError("modulated_deform_conv2d: with conv groups > 1 the output channels per group must be "
                                      "a multiple of 8")
        wt = _chk(weight.detach(), "weight")
        st = _stream(x)
        _launch("dcnv2_generic", 2.0 * cin_g * kh * kw * cout * n * ho * wo,
                4.0 * n * (cin * h * w + (3 * deform_groups * kh * kw + cout) * ho * wo), x,
                lambda: lib().eavsr_dcnv2_generic_f32(_p(x), _p(offset), _p(mask), _p(wt), _p(b), _p(out), n, cin, h, w, cout,
                                                      kh, kw, sh, sw, ph, pw, dh, dw, groups, deform_groups, st),
                "dcnv2_generic")
        return out
    st = _stream(x)
    px = float(n) * h * w
    flops, nbytes = 2.0 * cin * 9 * cout * px, 4.0 * px * (cin + 27 * deform_groups + cout)
    octets = (cin // deform_groups) // 8
    if DCN_MODE in ("il6", "il9") and octets & (octets - 1) == 0:     # the IL8 kernel wants a power-of-two number of octets per group
        # generic callers hold an NCHW tensor: one conversion pass to the IL8 layout, then the hot-path kernel (inside
        # MultiAdSTN the warp before the call writes IL8 itself and the predictor heads are passed instead of offset / mask)
        return dcnv2_il(to_il8(x), offset, mask, weight, b, deform_groups, nprod=int(DCN_MODE[2]), heads=False)
    if DCN_MODE == "bf16x9":
        require_lab("dcn mode 'bf16x9'")
    if DCN_MODE == "bf16x9" and w % 4 == 0 and x.data_ptr() % 16 == 0:
        wx = _packed_dcn_x9(weight)
        _launch("dcnv2_x9", flops, nbytes, x,
                lambda: lib().eavsr_dcnv2_f32x9(_p(x), _p(offset), _p(mask), _p(wx), _p(b), _p(out), n, cin, h, w, cout,
                                                deform_groups, st), "dcnv2_f32x9")
        return out
    wp = pack_cache.get([weight])
    _launch("dcnv2", flops, nbytes, x,
            lambda: lib().eavsr_dcnv2_f32(_p(x), _p(offset), _p(mask), _p(wp), _p(b), _p(out), n, cin, h, w, cout,
                                          deform_groups, st), "dcnv2")
    return out


def to_il8(x: Tensor) -> Tensor:
    """(n, c, h, w) fp32 -> "IL8" (n, c/8, h, w, 8): the 8 channels of a deformable group interleaved per pixel, the input
    layout of eavsr_dcnv2_il_f32"""
    x = _chk(x, "x")
    n, c, h, w = x.shape
    if c % 8:
        raise ValueError("to_il8: channels must be a multiple of 8")
    out = torch.empty((n, c // 8, h, w, 8), device=x.device, dtype=torch.float32)
    st = _stream(x)
    _launch("nchw_to_il8", 0.0, 8.0 * x.numel(), x,
            lambda: lib().eavsr_nchw_to_il8_f32(_p(x), _p(out), n, c, h, w, st), "nchw_to_il8")
    return out


def dcnv2_il(x_il8: Tensor, offset_or_heads: Tensor, mask: Optional[Tensor], weight: Tensor, bias: Optional[Tensor],
             deform_groups: int, nprod: int = 6, heads: bool = False, mask_activated: bool = False) -> Tensor:
    """DCNv2 (3x3, stride 1, pad 1) on an IL8 input.  heads=False: `offset_or_heads` / `mask` as mmcv's offset (n,18D,h,w)
    and mask (n,9D,h,w).  heads=True: `offset_or_heads` is the (n,15D,h,w) output of AdaptBlockOffset's three 5x5 heads and
    the kernel applies networks.py:302-315 (affine -> offsets, sigmoid) itself.  nprod: 9 = exact bf16x9, 6 = the three
    products below 2^-23 dropped."""
    x_il8 = _chk(x_il8, "x_il8")
    n, oct_, h, w, e = x_il8.shape
    if e != 8:
        raise ValueError("dcnv2_il: input must be IL8 (n, c/8, h, w, 8)")
    cin = oct_ * 8
    cout = int(weight.shape[0])
    if tuple(weight.shape[1:]) != (cin, 3, 3):
        raise ValueError("dcnv2_il: weight / input channel mismatch")
    oh = _chk(offset_or_heads, "offset_or_heads")
    D = int(deform_groups)
    if heads:
        if tuple(oh.shape) != (n, 15 * D, h, w):
            raise ValueError(f"dcnv2_il: heads shape {tuple(oh.shape)} != {(n, 15 * D, h, w)}")
        mask = None
    else:
        mask = _chk(mask, "mask")
        if tuple(oh.shape) != (n, 18 * D, h, w) or tuple(mask.shape) != (n, 9 * D, h, w):
            raise ValueError("dcnv2_il: offset / mask shape")
    b = None if bias is None else _chk(bias.detach(), "bias")
    if heads and _dcn_probe is not None:
        _dcn_probe.append(dcn_offset_stats(oh, D))
    impl = DCN_IL_IMPL
    if impl == "il2" and cin % 16:
        impl = "il"           # the round-4 schedule contracts groups in pairs
    if mask_activated and not (heads and impl == "il2"):
        raise ValueError("mask_activated: heads mode of the 'il2' schedule only (see heads_mask_activated())")
    wx = _packed_dcn_il2(weight) if impl == "il2" else _packed_dcn_x9(weight)
    out = torch.empty((n, cout, h, w), device=x_il8.device, dtype=torch.float32)
    st = _stream(out)
    px = float(n) * h * w
    # algorithmic bytes of the DCNv2 op as SURVEY 8d defines them (input + 27 D offset/mask + output); in heads mode the
    # kernel itself moves (cin + 15 D + cout) floats per pixel
    if impl == "ws":
        require_lab("the wave-specialised DCNv2 schedule 'ws'")
    fn = getattr(lib(), {"ws": "eavsr_dcnv2_ws_f32", "il": "eavsr_dcnv2_il_f32", "il2": "eavsr_dcnv2_il2_f32"}[impl])
    _launch("dcnv2_il" + ("_heads" if heads else ""), 2.0 * cin * 9 * cout * px, 4.0 * px * (cin + 27 * D + cout), out,
            lambda: fn(_p(x_il8), _p(oh), _p(mask), _p(wx), _p(b), _p(out), n, cin, h, w, cout, D,
                       int(nprod), (2 if mask_activated else 1) if heads else 0, st), "dcnv2_il")
    return out


def heads_mask_activated(cin: int, deform_groups: int = 8) -> bool:
    """True when the fused alignment should ask the predictor's heads convolution for MASKS (sigmoid in its epilogue,
    networks.py:313-314) rather than mask logits: the round-4 DCNv2 schedule takes them as they are (heads = 2), which moves four
    vector instructions per sample out of the kernel whose bound is vector issue.  The epilogue activates whole octets of output
    channels, so the first mask channel 6 D must be a multiple of 8 (D = 4, 8, ..; ADVICE r4: D = 1 / 2 used to raise)."""
    return (DCN_IL_IMPL == "il2" and cin % 16 == 0 and CONV5_MODE == "bf16x6" and HEADS_MASK_ACTIVATED
            and (6 * int(deform_groups)) % 8 == 0)


_dcn_probe = None      # bench.py: a list that collects offset statistics of every heads-mode DCNv2 call inside `dcn_probe()`


class dcn_probe:
    """Collect, per heads-mode DCNv2 call inside the scope, what the sampler saw: the offsets implied by the predictor heads
    (networks.py:302-315) and how many samples left the kernel's LDS window (served from global memory)."""

    def __enter__(self):
        global _dcn_probe
        _dcn_probe = []
        return _dcn_probe

    def __exit__(self, *exc):
        global _dcn_probe
        _dcn_probe = None
        return False


def dcn_offset_stats(heads: Tensor, D: int, tile_h: int = 8, tile_w: int = 32, margin_up: int = 6, margin_left: int = 8,
                     win_h: int = 20, win_w: int = 48) -> dict:
    """Statistics of the 18 D offsets that AdaptBlockOffset's heads imply (offset = T.R - R + t per deformable group) and the
    fraction of (pixel, tap) samples whose bilinear corners leave the DCNv2 kernel's LDS window (tile 8 x 32, window rows
    y0-6 .. y0+13, columns x0-8 .. x0+39: csrc/dcnv2_il2.hip).  Plain torch on the tensor's device; measurement only."""
    n, _, h, w = heads.shape
    T = heads[:, :4 * D].view(n, D, 2, 2, h, w)
    t = heads[:, 4 * D:6 * D].view(n, D, 2, 1, h, w)
    ky = torch.tensor([-1., -1., -1., 0., 0., 0., 1., 1., 1.], device=heads.device)
    kx = torch.tensor([-1., 0., 1., -1., 0., 1., -1., 0., 1.], device=heads.device)
    R = torch.stack([ky, kx], 0).view(1, 1, 2, 9, 1, 1)
    off = (T[:, :, :, 0:1] * R[:, :, 0:1] + T[:, :, :, 1:2] * R[:, :, 1:2]) - R + t      # (n, D, 2, 9, h, w): dy, dx per tap
    dy, dx = off[:, :, 0], off[:, :, 1]
    ys = torch.arange(h, device=heads.device).view(1, 1, 1, h, 1) % tile_h
    xs = torch.arange(w, device=heads.device).view(1, 1, 1, 1, w) % tile_w
    ry = margin_up + ys + ky.view(1, 1, 9, 1, 1) + torch.floor(dy)
    rx = margin_left + xs + kx.view(1, 1, 9, 1, 1) + torch.floor(dx)
    outside = (ry < 0) | (ry > win_h - 2) | (rx < 0) | (rx > win_w - 2)
    mag = torch.sqrt(dy * dy + dx * dx)
    return {"mean_abs_dy": float(dy.abs().mean()), "mean_abs_dx": float(dx.abs().mean()), "mean_norm": float(mag.mean()),
            "sigma_dy": float(dy.std()), "sigma_dx": float(dx.std()), "max_abs": float(torch.maximum(dy.abs().max(), dx.abs().max())),
            "frac_outside_lds_window": float(outside.float().mean())}


# Schedule of the IL8 DCNv2 kernel (same arithmetic, same arguments): "il" = eavsr_dcnv2_il_f32, every wave samples and
# contracts (default: faster at the alignment's offsets, |offset| of a few pixels); "ws" = eavsr_dcnv2_ws_f32, wave-specialised
# samplers / contractors (csrc/dcnv2_ws.hip): ~18 % slower there, 8-15 % faster when many samples leave the LDS window
# (sigma = 4 px), because its out-of-window samples are blended in line instead of in a second MFMA pass.  DESIGN.md 4b.
# "il2" = eavsr_dcnv2_il2_f32 (csrc/dcnv2_il2.hip, round 4): the two groups of a pair as nine full k-steps in one software
# pipeline across group / pair / tile boundaries (same operands and products; falls back to "il" when cin % 16 != 0).
DCN_IL_IMPL = os.environ.get("EAVSR_DCN_IL_IMPL", "il2")
HEADS_MASK_ACTIVATED = os.environ.get("EAVSR_HEADS_SIGMOID", "1") != "0"      # A/B switch of heads_mask_activated()


def set_dcn_il_impl(impl: str) -> None:
    global DCN_IL_IMPL
    if impl not in ("il", "ws", "il2"):
        raise ValueError(f"dcn il impl {impl!r}: 'il', 'il2' or 'ws'")
    if impl == "ws":
        require_lab("the wave-specialised DCNv2 schedule 'ws'")
    DCN_IL_IMPL = impl


_il2_pack_cache = register_weight_cache({})


def _packed_dcn_il2(weight: Tensor) -> Tensor:
    """Pre-split (3 x bf16) weight slab of a 3x3 DCNv2 weight in the pair-step order of eavsr_dcnv2_il2_f32; cached per weight
    object and version, verified by identity."""
    key = (id(weight), weight._version)
    hit = _il2_pack_cache.get(key)
    if hit is not None and hit[0]() is weight:
        return hit[1]
    w = _chk(weight.detach(), "weight")
    cout, cin = int(w.shape[0]), int(w.shape[1])
    nbytes = lib().eavsr_dcn_weight_il2_bytes(cout, cin)
    if nbytes <= 0 or tuple(w.shape[2:]) != (3, 3):
        raise NotImplementedError(f"il2 weight shape {tuple(w.shape)} unsupported")
    packed = torch.empty(nbytes // 4, device=w.device, dtype=torch.int32)
    with _DeviceOf(w):
        N.check(lib().eavsr_pack_dcn_weight_il2(_p(w), _p(packed), cout, cin, _stream(w)), "pack_dcn_weight_il2")
    for k in [k for k in _il2_pack_cache if k[0] == id(weight)]:
        _il2_pack_cache.pop(k, None)
    _il2_pack_cache[key] = (weakref.ref(weight, lambda _r, k=key, c=_il2_pack_cache: c.pop(k, None)), packed)
    return packed


_il16_pack_cache = register_weight_cache({})


def _packed_il16(weight: Tensor, code: int) -> Tensor:
    key = (id(weight), weight._version, code)
    hit = _il16_pack_cache.get(key)
    if hit is not None and hit[0]() is weight:
        return hit[1]
    w = _chk(weight.detach(), "weight")
    cout, cin = int(w.shape[0]), int(w.shape[1])
    if tuple(w.shape[2:]) != (3, 3):
        raise NotImplementedError("dcnv2_il16 is 3x3")
    nbytes = lib().eavsr_dcn_il16_weight_bytes(cout, cin)
    if nbytes <= 0:
        raise NotImplementedError(f"dcnv2_il16 weight shape {tuple(w.shape)} unsupported")
    packed = torch.empty(nbytes // 4, device=w.device, dtype=torch.int32)
    with _DeviceOf(w):
        N.check(lib().eavsr_pack_dcn_il16_weight(_p(w), _p(packed), cout, cin, code, _stream(w)), "pack_dcn_il16_weight")
    for k in [k for k in _il16_pack_cache if k[0] == id(weight)]:
        _il16_pack_cache.pop(k, None)
    _il16_pack_cache[key] = (weakref.ref(weight, lambda _r, k=key, c=_il16_pack_cache: c.pop(k, None)), packed)
    return packed


def to_il8_h16(x: Tensor, dtype) -> Tensor:
    """(n, c, h, w) fp32 -> IL8 (n, c/8, h, w, 8) rounded to 'fp16' / 'bf16'"""
    x = _chk(x, "x")
    n, c, h, w = x.shape
    if c % 8:
        raise ValueError("to_il8_h16: channels must be a multiple of 8")
    code = h16_code(dtype)
    out = torch.empty((n, c // 8, h, w, 8), device=x.device, dtype=_H16_TORCH[code])
    st = _stream(x)
    _launch("nchw_to_il8_h16", 0.0, 6.0 * x.numel(), x,
            lambda: lib().eavsr_nchw_to_il8_h16(_p(x), _p(out), n, c, h, w, code, st), "nchw_to_il8_h16")
    return out


def dcnv2_il16(x_il8: Tensor, offset_or_heads: Tensor, mask: Optional[Tensor], weight: Tensor, bias: Optional[Tensor],
               deform_groups: int, heads: bool = False) -> Tensor:
    """DCNv2 (3x3, stride 1, pad 1) on a 16-bit IL8 input (bf16 / fp16 samples and weights, fp32 blend and accumulation,
    fp32 output); arguments as `dcnv2_il`."""
    x_il8 = _chk_h16(x_il8, "x_il8")
    n, oct_, h, w, e = x_il8.shape
    if e != 8:
        raise ValueError("dcnv2_il16: input must be IL8 (n, c/8, h, w, 8)")
    code = h16_code(x_il8.dtype)
    cin, cout, D = oct_ * 8, int(weight.shape[0]), int(deform_groups)
    if tuple(weight.shape[1:]) != (cin, 3, 3):
        raise ValueError("dcnv2_il16: weight / input channel mismatch")
    oh = _chk(offset_or_heads, "offset_or_heads")
    if heads:
        if tuple(oh.shape) != (n, 15 * D, h, w):
            raise ValueError(f"dcnv2_il16: heads shape {tuple(oh.shape)} != {(n, 15 * D, h, w)}")
        mask = None
    else:
        mask = _chk(mask, "mask")
        if tuple(oh.shape) != (n, 18 * D, h, w) or tuple(mask.shape) != (n, 9 * D, h, w):
            raise ValueError("dcnv2_il16: offset / mask shape")
    b = None if bias is None else _chk(bias.detach(), "bias")
    wp = _packed_il16(weight, code)
    out = torch.empty((n, cout, h, w), device=x_il8.device, dtype=torch.float32)
    st = _stream(out)
    px = float(n) * h * w
    # algorithmic bytes of the 16-bit DCNv2 as SURVEY 8a counts them: 344 elements x 2 bytes per pixel
    _launch("dcnv2_il16" + ("_heads" if heads else ""), 2.0 * cin * 9 * cout * px, 2.0 * px * (cin + 27 * D + cout), out,
            lambda: lib().eavsr_dcnv2_il16(_p(x_il8), _p(oh), _p(mask), _p(wp), _p(b), _p(out), n, cin, h, w, cout, D, code,
                                           1 if heads else 0, st), "dcnv2_il16")
    return out


# How DCNv2 runs (fp32 tensors in and out in every mode):
#   "il6" (default) = the round-2 hot-path kernel eavsr_dcnv2_il_f32: IL8 input layout, both fp32 operands split exactly into
#              three bf16 terms, the six partial products that matter on the bf16 MFMA, fp32 accumulation (the three dropped
#              products are < 2^-23 of the result each: one fp32 rounding).  Inside MultiAdSTN (inference) it is fed by the
#              paired warp (IL8 output) and by the predictor's head channels directly (affine -> offsets and the mask sigmoid
#              happen in the kernel); a plain modulated_deform_conv2d call converts its NCHW input first.
#   "il9"    = the same kernel with all nine partial products (exact operands, only the fp32 accumulation rounds)
#   "native" = round 1's LDS-window kernel on v_mfma_f32_32x32x2_f32 (an fp32 fma chain)
#   "bf16x9" = round 1's bf16x9 kernel (NCHW input, LDS window per channel plane)
DCN_MODE = os.environ.get("EAVSR_DCN_MODE", "il6")


def set_dcn_mode(mode: str) -> None:
    global DCN_MODE
    if mode not in ("native", "bf16x9", "il6", "il9"):
        raise ValueError(f"dcn mode {mode!r}: 'native', 'bf16x9', 'il6' or 'il9'")
    if mode == "bf16x9":
        require_lab("dcn mode 'bf16x9' (the round-1 NCHW kernel with nine partial products)")
    DCN_MODE = mode


_x9_pack_cache = register_weight_cache({})


def _packed_x9(weights: Sequence[Tensor]) -> Tensor:
    """Pre-split (3 x bf16) weight slab of a 3x3 weight -- or of several stacked along cout -- for the bf16x9 kernels;
    cached per weight objects and versions, verified by identity (as pack_cache)."""
    key = tuple((id(w), w._version) for w in weights)
    hit = _x9_pack_cache.get(key)
    if hit is not None and all(r() is w for r, w in zip(hit[0], weights)):
        return hit[1]
    w = weights[0] if len(weights) == 1 else torch.cat([x.detach() for x in weights], 0)
    w = _chk(w.detach(), "weight")
    cout, cin = int(w.shape[0]), int(w.shape[1])
    if tuple(w.shape[2:]) != (3, 3):
        raise NotImplementedError("bf16x9 kernels are 3x3")
    nbytes = lib().eavsr_dcn_weight_x9_bytes(cout, cin)
    if nbytes <= 0:
        raise NotImplementedError(f"bf16x9 weight shape {tuple(w.shape)} unsupported")
    packed = torch.empty(nbytes // 4, device=w.device, dtype=torch.int32)
    with _DeviceOf(w):
        N.check(lib().eavsr_pack_dcn_weight_x9(_p(w), _p(packed), cout, cin, _stream(w)), "pack_dcn_weight_x9")
    ids = {id(x) for x in weights}
    for k in [k for k in _x9_pack_cache if any(i in ids for i, _ in k)]:
        _x9_pack_cache.pop(k, None)
    refs = tuple(weakref.ref(x, lambda _r, k=key, c=_x9_pack_cache: c.pop(k, None)) for x in weights)
    _x9_pack_cache[key] = (refs, packed)
    return packed


def _packed_dcn_x9(weight: Tensor) -> Tensor:
    return _packed_x9([weight])


# How the large 3x3 convolutions are computed (all fp32 in, fp32 out, fp32 accumulation):
#   "winograd" = Winograd F(2x2, 3x3) on v_mfma_f32_16x16x4_f32 (eavsr_conv3x3_wino_f32): 2.25x fewer
#                multiplications, what cuDNN / MIOpen run for fp32 3x3 convolutions; measured closer to fp64 than the
#                direct sum (tests/test_hip_ops.py)
#   "winograd4" (default) = Winograd F(4x4, 3x3) (eavsr_conv3x3_wino4_f32): 4x fewer multiplications, ~1e-5 relative
#                error per convolution instead of 4e-7 (end to end on the bench clip: 2.4e-7 against 1.8e-7, the path's
#                bound is 1e-3); falls back to F(2x2, 3x3) for the fused channel-attention prologue
#   "direct"   = direct sum on v_mfma_f32_32x32x2_f32 (eavsr_conv2d_f32) -- also what every shape the Winograd kernel
#                does not cover runs (small images, widths not a multiple of 4, channels not a multiple of 8, 1x1/5x5/7x7)
#   "bf16x9"   = direct sum with an exact three-way bf16 split of both operands, nine partial products in fp32
#                (eavsr_conv3x3_f32x9), opt-in
def _norm_conv_mode(mode: str) -> str:
    mode = {"native": "direct"}.get(mode, mode)
    if mode not in ("direct", "winograd", "winograd4", "bf16x9"):
        raise ValueError(f"conv mode {mode!r}: 'winograd', 'winograd4', 'direct' or 'bf16x9'")
    return mode


CONV_MODE = _norm_conv_mode(os.environ.get("EAVSR_CONV_MODE", "winograd4"))
# 3x3 64 -> cout launches of at most X6S_MAX_TILES 8 x 32-pixel tiles (where the fp32-MFMA small kernel's time is one workgroup's
# chain of matrix instructions): "x6s" = csrc/conv3_x6s.hip (exact bf16x6), "direct" = the fp32 MFMA kernels (A/B switch)
CONV3_SMALL = os.environ.get("EAVSR_CONV3_SMALL", "x6s")
X6S_MAX_TILES = int(os.environ.get("EAVSR_X6S_MAX_TILES", "256"))
WINO_MIN_TILES = int(os.environ.get("EAVSR_WINO_MIN_TILES", "192"))   # 8 x 32-px tiles per launch below which the direct kernel runs


def set_conv_mode(mode: str) -> None:
    global CONV_MODE
    mode = _norm_conv_mode(mode)
    if mode in ("winograd", "bf16x9"):      # F(2x2,3x3) and the nine-product direct kernel: retired schedules
        require_lab(f"conv mode {mode!r}")
    CONV_MODE = mode


@contextlib.contextmanager
def modes(conv: Optional[str] = None, dcn: Optional[str] = None, dcn_il_impl: Optional[str] = None):
    """Scoped kernel selection: `with ops.modes(conv="direct", dcn="native"): ...` switches the process-wide defaults for the
    body and restores them on exit (also on an exception), so that a caller -- a test, an A/B measurement -- never leaves another
    mode behind.  The switches select between implementations of the same arithmetic; they are process-wide (one forward runs one
    kernel set), not per call."""
    prev = (CONV_MODE, DCN_MODE, DCN_IL_IMPL)
    try:
        if conv is not None:
            set_conv_mode(conv)
        if dcn is not None:
            set_dcn_mode(dcn)
        if dcn_il_impl is not None:
            set_dcn_il_impl(dcn_il_impl)
        yield
    finally:
        set_conv_mode(prev[0])
        set_dcn_mode(prev[1])
        set_dcn_il_impl(prev[2])


_wino_pack_cache = register_weight_cache({})


def _packed_wino(weights: Sequence[Tensor], four: bool = False, kind: Optional[str] = None) -> Tensor:
    """G g G^T of a conv weight (or of several stacked along cout); kind "f2" = F(2x2, 3x3), "f4" = F(4x4, 3x3),
    "f5" = F(2x2, 5x5); cached per weight objects and versions."""
    kind = kind or ("f4" if four else "f2")
    if kind == "f2":
        require_lab("Winograd F(2x2,3x3)")
    key = tuple((id(w), w._version) for w in weights) + ((kind, 0),)
    hit = _wino_pack_cache.get(key)
    if hit is not None and all(r() is w for r, w in zip(hit[0], weights)):
        return hit[1]
    w = weights[0] if len(weights) == 1 else torch.cat([x.detach() for x in weights], 0)
    w = _chk(w.detach(), "weight")
    cout, cin = int(w.shape[0]), int(w.shape[1])
    elems = (lib().eavsr_wino_weight_elems if kind == "f2" else lib().eavsr_wino4_weight_elems)(cout, cin)
    if elems <= 0 or tuple(w.shape[2:]) != ((5, 5) if kind == "f5" else (3, 3)):
        raise NotImplementedError(f"winograd weight shape {tuple(w.shape)} unsupported")
    packed = torch.empty(elems, device=w.device, dtype=torch.float32)
    with _DeviceOf(w):
        fn = getattr(lib(), {"f2": "eavsr_pack_conv_weight_wino", "f4": "eavsr_pack_conv_weight_wino4",
                             "f5": "eavsr_pack_conv_weight_wino5x5"}[kind])
        N.check(fn(_p(w), _p(packed), cout, cin, _stream(w)), "pack_conv_weight_wino")
    ids = {id(x) for x in weights}
    for k in [k for k in _wino_pack_cache if k[-1] == (kind, 0) and any(i in ids for i, _ in k[:-1])]:
        _wino_pack_cache.pop(k, None)
    refs = tuple(weakref.ref(x, lambda _r, k=key, c=_wino_pack_cache: c.pop(k, None)) for x in weights)
    _wino_pack_cache[key] = (refs, packed)
    return packed


# ------------------------------------------------------------------------------------------
# predictor pieces
# ------------------------------------------------------------------------------------------
def adapt_frontend(x: Tensor, h_hr: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor) -> Tensor:
    x, h_hr = _chk(x, "x"), _chk(h_hr, "h_hr")
    if x.shape != h_hr.shape:
        raise ValueError("x and h_hr must have the same shape")
    n, c, h, w = x.shape
    if tuple(w1.shape) != (2 * c, 1, 3, 3) or tuple(w2.shape) != (c, 2, 3, 3):
        raise ValueError("adapt_frontend weight shapes")
    out = torch.empty_like(x)
    w1, b1, w2, b2 = (_chk(v.detach(), "param") for v in (w1, b1, w2, b2))
    st = _stream(x)
    _launch("adapt_frontend", 2.0 * 27 * c * n * h * w, 4.0 * 3 * c * n * h * w, x,
            lambda: lib().eavsr_adapt_frontend_f32(_p(x), _p(h_hr), _p(w1), _p(b1), _p(w2), _p(b2), _p(out),
                                                   n, c, h, w, st), "adapt_frontend")
    return out


def flow_level(x: Tensor, h_hr: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, head_ws: Sequence[Tensor],
               head_bs: Sequence[Tensor], wt: Tensor, bt: Tensor) -> Tensor:
    """TransOffsetworelu(AdaptBlock2_3x3(x, h_hr)) as one kernel (networks.py:334-348 + 566-571): (n,c,h,w) x 2 -> (n,2,h,w).
    head_ws / head_bs: [transform_matrix_conv, translation_conv] parameters (4 + 2 output channels)."""
    require_lab("ops.flow_level (one kernel per pyramid level)")
    x, h_hr = _chk(x, "x"), _chk(h_hr, "h_hr")
    if x.shape != h_hr.shape:
        raise ValueError("x and h_hr must have the same shape")
    n, c, h, w = x.shape
    wh, bh = _cat_weights(list(head_ws)), _bias_of(list(head_bs))
    if tuple(w1.shape) != (2 * c, 1, 3, 3) or tuple(w2.shape) != (c, 2, 3, 3) or tuple(wh.shape) != (6, c, 3, 3) or \
            tuple(wt.shape) != (2, 18, 3, 3) or bh is None:
        raise ValueError("flow_level: parameter shapes")
    w1, b1, w2, b2, wt, bt = (_chk(v.detach(), "param") for v in (w1, b1, w2, b2, wt, bt))
    out = torch.empty((n, 2, h, w), device=x.device, dtype=torch.float32)
    st = _stream(x)
    px = float(n) * h * w
    _launch("flow_level", 2.0 * px * (27 * c + 54 * c + 324), 4.0 * px * (2 * c + 2), x,
            lambda: lib().eavsr_flow_level_f32(_p(x), _p(h_hr), _p(w1), _p(b1), _p(w2), _p(b2), _p(wh), _p(bh), _p(wt), _p(bt),
                                               _p(out), n, c, h, w, st), "flow_level")
    return out


def affine_offsets(heads: Tensor, D: int, with_mask: bool) -> Tuple[Tensor, Optional[Tensor]]:
    heads = _chk(heads, "heads")
    n, hc, h, w = heads.shape
    if hc != (15 * D if with_mask else 6 * D):
        raise ValueError(f"heads has {hc} channels, expected {15 * D if with_mask else 6 * D}")
    off = torch.empty((n, 18 * D, h, w), device=heads.device, dtype=torch.float32)
    mask = torch.empty((n, 9 * D, h, w), device=heads.device, dtype=torch.float32) if with_mask else None
    st = _stream(heads)
    _launch("affine_offsets", 60.0 * D * n * h * w, 4.0 * n * h * w * (hc + 18 * D + (9 * D if with_mask else 0)), heads,
            lambda: lib().eavsr_affine_offsets_f32(_p(heads), _p(off), _p(mask), n, D, h, w, st), "affine_offsets")
    return off, mask


# ------------------------------------------------------------------------------------------
# resampling glue
# ------------------------------------------------------------------------------------------
def resize_bilinear_ac(x: Tensor, size: Tuple[int, int], scale: float = 1.0, pre_add: Optional[Tensor] = None,
                       post_add: Optional[Tensor] = None) -> Tensor:
    """scale * F.interpolate(x [+ pre_add], size, bilinear, align_corners=True) [+ post_add]"""
    x = _chk(x, "x")
    n, c, hin, win = x.shape
    hout, wout = int(size[0]), int(size[1])
    if pre_add is not None:
        pre_add = _chk(pre_add, "pre_add")
        if pre_add.shape != x.shape:
            raise ValueError("pre_add shape")
    out = torch.empty((n, c, hout, wout), device=x.device, dtype=torch.float32)
    if post_add is not None:
        post_add = _chk(post_add, "post_add")
        if post_add.shape != out.shape:
            raise ValueError("post_add shape")
    st = _stream(x)
    _launch("resize_bilinear_ac", 0.0, 4.0 * n * c * (hin * win + hout * wout), x,
            lambda: lib().eavsr_resize_bilinear_ac_f32(_p(x), _p(pre_add), _p(post_add), _p(out), n, c, hin, win, hout,
                                                       wout, float(scale), st), "resize_bilinear_ac")
    return out


def pyramid(x: Tensor) -> Tuple[Tensor, Tensor]:
    """(down2, down4) of eavsrp_model.py:218-220"""
    x = _chk(x, "x")
    n, c, h, w = x.shape
    if h % 4 or w % 4:
        raise ValueError(f"h={h}, w={w} must be divisible by 4 (eavsrp_model.py:223-224)")
    d2 = torch.empty((n, c, h // 2, w // 2), device=x.device, dtype=torch.float32)
    d4 = torch.empty((n, c, h // 4, w // 4), device=x.device, dtype=torch.float32)
    st = _stream(x)
    _launch("pyramid", 0.0, 4.0 * n * c * h * w * (1 + 0.25 + 0.0625), x,
            lambda: lib().eavsr_pyramid_f32(_p(x), _p(d2), _p(d4), n * c, h, w, st), "pyramid")
    return d2, d4


def add(a: Tensor, b: Tensor, c: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """a + b (+ c); `out`: an existing contiguous fp32 tensor of the same shape to write into (e.g. one frame's slice of a
    frame-major buffer) instead of a new one"""
    a, b = _chk(a, "a"), _chk(b, "b")
    if a.shape != b.shape or (c is not None and c.shape != a.shape):
        raise ValueError("add: shape mismatch")
    if c is not None:
        c = _chk(c, "c")
    if out is None:
        out = torch.empty_like(a)
    elif out.shape != a.shape or out.dtype != torch.float32 or not out.is_cuda or not out.is_contiguous():
        raise ValueError("add: `out` must be a contiguous fp32 GPU tensor of the operands' shape")
    st = _stream(a)
    _launch("add", float(a.numel()), 4.0 * a.numel() * (3 if c is None else 4), a,
            lambda: lib().eavsr_add_f32(_p(a), _p(b), _p(c), _p(out), a.numel(), st), "add")
    return out


def normalize(x: Tensor, mean: Tensor, std: Tensor) -> Tensor:
    """(x - mean) / std per channel (models/eavsrp_model.py:436-437, networks.py:550); mean / std: c floats on the device"""
    x = _chk(x, "x")
    n, c, h, w = x.shape
    mean, std = _chk(mean.reshape(-1), "mean"), _chk(std.reshape(-1), "std")
    if mean.numel() != c or std.numel() != c:
        raise ValueError("normalize: one mean / std per channel")
    out = torch.empty_like(x)
    st = _stream(x)
    _launch("normalize", 2.0 * x.numel(), 8.0 * x.numel(), x,
            lambda: lib().eavsr_normalize_f32(_p(x), _p(mean), _p(std), _p(out), n, c, h * w, st), "normalize")
    return out


def avg_pool2(x: Tensor) -> Tensor:
    """F.avg_pool2d(x, 2, 2, count_include_pad=False) for even h, w (models/eavsrp_model.py:450-462)"""
    x = _chk(x, "x")
    n, c, h, w = x.shape
    if h % 2 or w % 2:
        raise ValueError(f"avg_pool2: h={h}, w={w} must be even")
    out = torch.empty((n, c, h // 2, w // 2), device=x.device, dtype=torch.float32)
    st = _stream(x)
    _launch("avg_pool2", 4.0 * out.numel(), 4.0 * (x.numel() + out.numel()), x,
            lambda: lib().eavsr_avg_pool2_f32(_p(x), _p(out), n * c, h, w, st), "avg_pool2")
    return out


def resize_bilinear(x: Tensor, size: Tuple[int, int], channel_mul: Optional[Tuple[float, float]] = None) -> Tensor:
    """F.interpolate(x, size, mode='bilinear', align_corners=False); `channel_mul` = (m0, m1) multiplies channels 0 / 1 of the
    result (the flow rescaling of models/eavsrp_model.py:519-521)"""
    x = _chk(x, "x")
    n, c, hin, win = x.shape
    hout, wout = int(size[0]), int(size[1])
    out = torch.empty((n, c, hout, wout), device=x.device, dtype=torch.float32)
    m0, m1 = (1.0, 1.0) if channel_mul is None else (float(channel_mul[0]), float(channel_mul[1]))
    st = _stream(x)
    _launch("resize_bilinear", 8.0 * out.numel(), 4.0 * (x.numel() + out.numel()), x,
            lambda: lib().eavsr_resize_bilinear_f32(_p(x), _p(out), n, c, hin, win, hout, wout, 0 if channel_mul is None else c,
                                                    m0, m1, st), "resize_bilinear")
    return out


def concat3(a: Tensor, b: Tensor, c: Tensor) -> Tensor:
    """torch.cat([a, b, c], 1) (models/eavsrp_model.py:486)"""
    a, b, c = _chk(a, "a"), _chk(b, "b"), _chk(c, "c")
    n, ca, h, w = a.shape
    if b.shape[0] != n or c.shape[0] != n or b.shape[2:] != a.shape[2:] or c.shape[2:] != a.shape[2:]:
        raise ValueError("concat3: shape mismatch")
    cb, cc = b.shape[1], c.shape[1]
    out = torch.empty((n, ca + cb + cc, h, w), device=a.device, dtype=torch.float32)
    st = _stream(a)
    _launch("concat3", 0.0, 8.0 * out.numel(), a,
            lambda: lib().eavsr_concat3_f32(_p(a), ca, _p(b), cb, _p(c), cc, _p(out), n, h * w, st), "concat3")
    return out


# ------------------------------------------------------------------------------------------
# channel attention
# ------------------------------------------------------------------------------------------
def ca_scale(partial: Tensor, hw: int, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, with_mean: bool = False):
    """partial (n, tiles, c) channel sums -> sigmoid(W2 relu(W1 mean + b1) + b2), shape (n, c); with_mean: also the means (n, c)"""
    partial = _chk(partial, "partial")
    n, tiles, c = partial.shape
    cr = int(w1.shape[0])
    scale = torch.empty((n, c), device=partial.device, dtype=torch.float32)
    mean = torch.empty((n, c), device=partial.device, dtype=torch.float32) if with_mean else None
    w1, b1, w2, b2 = (_chk(v.detach(), "param") for v in (w1, b1, w2, b2))
    st = _stream(partial)
    _launch("ca_scale", 0.0, 4.0 * partial.numel(), partial,
            lambda: lib().eavsr_ca_scale_mean_f32(_p(partial), tiles, hw, _p(w1), _p(b1), _p(w2), _p(b2), _p(scale), _p(mean), n, c,
                                                  cr, st), "ca_scale")
    return (scale, mean) if with_mean else scale


def ca_tail(r: Tensor, partial: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, x: Tensor, with_stats: bool = False):
    """the RCAB tail in one launch: r * sigmoid(W2 relu(W1 mean_hw(r) + b1) + b2) + x with mean_hw(r) from the conv's per-tile
    channel sums `partial` (n, tiles, c); falls back to ca_scale + scale_residual where the fused kernel's alignment rules fail.
    with_stats=True also returns the attention (n, c) and the channel means (n, c) (the backward of CALayer needs them):
    (out, scale, mean)."""
    r, x, partial = _chk(r, "r"), _chk(x, "x"), _chk(partial, "partial")
    n, c, h, w = r.shape
    if x.shape != r.shape or partial.shape[0] != n or partial.shape[2] != c:
        raise ValueError("ca_tail: shape mismatch")
    hw = h * w
    if hw % 4 or (r.data_ptr() | x.data_ptr()) % 16 or c > 256:
        if with_stats:
            scale, mean = ca_scale(partial, hw, w1, b1, w2, b2, with_mean=True)
            return scale_residual(r, scale, x), scale, mean
        return scale_residual(r, ca_scale(partial, hw, w1, b1, w2, b2), x)
    tiles = int(partial.shape[1])
    cr = int(w1.shape[0])
    w1, b1, w2, b2 = (_chk(v.detach(), "param") for v in (w1, b1, w2, b2))
    out = torch.empty_like(r)
    scale = torch.empty((n, c), device=r.device, dtype=torch.float32) if with_stats else None
    mean = torch.empty((n, c), device=r.device, dtype=torch.float32) if with_stats else None
    st = _stream(r)
    _launch("ca_tail", 2.0 * r.numel(), 12.0 * r.numel(), r,
            lambda: lib().eavsr_ca_tail_stats_f32(_p(r), _p(partial), tiles, _p(w1), _p(b1), _p(w2), _p(b2), _p(x), _p(out),
                                                  _p(scale), _p(mean), n, c, cr, hw, st), "ca_tail")
    return (out, scale, mean) if with_stats else out


def scale_residual(r: Tensor, scale: Tensor, x: Tensor) -> Tensor:
    """r * scale[n,c] + x"""
    r, x, scale = _chk(r, "r"), _chk(x, "x"), _chk(scale, "scale")
    n, c, h, w = r.shape
    if x.shape != r.shape or tuple(scale.shape) != (n, c):
        raise ValueError("scale_residual: shape mismatch")
    out = torch.empty_like(r)
    st = _stream(r)
    _launch("scale_residual", 2.0 * r.numel(), 12.0 * r.numel(), r,
            lambda: lib().eavsr_scale_residual_f32(_p(r), _p(scale), _p(x), _p(out), n, c, h * w, st), "scale_residual")
    return out


# ==========================================================================================
# backward kernels (training step; used by eavsr_amd/autograd.py)
# ==========================================================================================
def act_bwd(dy: Tensor, y: Tensor, act: str, slope: float = 0.0) -> Tensor:
    dy, y = _chk(dy, "dy"), _chk(y, "y")
    g = torch.empty_like(dy)
    st = _stream(dy)
    _launch("act_bwd", float(dy.numel()), 12.0 * dy.numel(), dy,
            lambda: lib().eavsr_act_bwd_f32(_p(dy), _p(y), _p(g), dy.numel(), ACT[act], float(slope), st), "act_bwd")
    return g


def plane_sum(a: Tensor, b: Optional[Tensor] = None, scale: float = 1.0) -> Tensor:
    """(n,c,h,w) -> (n,c): scale * sum_hw a (* b)"""
    a = _chk(a, "a")
    n, c, h, w = a.shape
    if b is not None:
        b = _chk(b, "b")
    out = torch.empty((n, c), device=a.device, dtype=torch.float32)
    st = _stream(a)
    _launch("plane_sum", float(a.numel()), 4.0 * a.numel() * (2 if b is not None else 1), a,
            lambda: lib().eavsr_plane_sum_f32(_p(a), _p(b), _p(out), n * c, h * w, float(scale), st), "plane_sum")
    return out


def channel_sum(a: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """(n,c,h,w) -> (c,): sum over batch and plane (bias gradient); `out` (+)= when given"""
    a = _chk(a, "a")
    n, c, h, w = a.shape
    if out is None:
        out = torch.empty((c,), device=a.device, dtype=torch.float32)
        accumulate = False
    elif tuple(out.shape) != (c,) or not out.is_contiguous() or out.dtype != torch.float32 or out.device != a.device:
        raise ValueError("channel_sum: out must be a contiguous fp32 (c,) tensor on the input's device")
    if n == 0 or c == 0:
        return out if accumulate else out.zero_()
    st = _stream(a)
    _launch("channel_sum", float(a.numel()), 4.0 * a.numel(), a,
            lambda: lib().eavsr_channel_sum_f32(_p(a), _p(out), n, c, h * w, int(accumulate), st), "channel_sum")
    return out


def scale_residual_bwd(d: Tensor, scale: Tensor, dmean: Optional[Tensor]) -> Tensor:
    d, scale = _chk(d, "d"), _chk(scale, "scale")
    n, c, h, w = d.shape
    if dmean is not None:
        dmean = _chk(dmean, "dmean")
    dr = torch.empty_like(d)
    st = _stream(d)
    _launch("scale_residual_bwd", float(d.numel()), 8.0 * d.numel(), d,
            lambda: lib().eavsr_scale_residual_bwd_f32(_p(d), _p(scale), _p(dmean), _p(dr), n, c, h * w, st),
            "scale_residual_bwd")
    return dr


def ca_mlp_bwd(mean: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, dscale: Tensor):
    mean, dscale = _chk(mean, "mean"), _chk(dscale, "dscale")
    w1, b1, w2, b2 = (_chk(v.detach(), "param") for v in (w1, b1, w2, b2))
    n, c = mean.shape
    cr = int(w1.shape[0])
    dmean = torch.empty_like(mean)
    dw1, db1, dw2, db2 = torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w2), torch.empty_like(b2)
    st = _stream(mean)
    _launch("ca_mlp_bwd", 0.0, 0.0, mean,
            lambda: lib().eavsr_ca_mlp_bwd_f32(_p(mean), _p(w1), _p(b1), _p(w2), _p(b2), _p(dscale), _p(dmean), _p(dw1),
                                               _p(db1), _p(dw2), _p(db2), n, c, cr, st), "ca_mlp_bwd")
    return dmean, dw1, db1, dw2, db2


def rcab_tail_bwd_supported(c: int, cr: int) -> bool:
    return c == 64 and cr in (1, 2, 4, 8)


def rcab_tail_bwd(d: Tensor, r: Tensor, mean: Tensor, scale: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor,
                  grads=None, accumulate: bool = False, dsum: Optional[Tensor] = None):
    """Backward of out = r * sigmoid(W2 relu(W1 mean_hw(r) + b1) + b2) + x w.r.t. r and the four MLP parameters: the plane
    sums sum_hw d r, then ONE launch (eavsr_rcab_tail_bwd_f32) for the MLP's backward, the broadcast of the mean's gradient and
    dr = d * scale + dmean / hw.  `grads` = (dw1, db1, dw2, db2) buffers to write or, with `accumulate`, to add to; fresh ones
    when None.  dsum: (n, rows, c) partial sums of d * r that the launch adds up itself (conv2d(.., dgrad=True, sum_mul=r) where d was
    produced) -- then there is no plane-sum launch.  Returns (dr, dw1, db1, dw2, db2)."""
    d, r, mean, scale = _chk(d, "d"), _chk(r, "r"), _chk(mean, "mean"), _chk(scale, "scale")
    w1, b1, w2, b2 = (_chk(v.detach(), "param") for v in (w1, b1, w2, b2))
    n, c, h, w = d.shape
    cr = int(w1.shape[0])
    if not rcab_tail_bwd_supported(c, cr):
        raise NotImplementedError(f"rcab_tail_bwd: {c} channels / {cr} hidden units")
    # `mean` (n, c), or the conv epilogue's partial channel sums (n, rows, c) that the kernel adds up itself
    if mean.dim() == 3:
        if tuple(mean.shape[::2]) != (n, c):
            raise ValueError("rcab_tail_bwd: partial sums must be (n, rows, c)")
        mean_rows = int(mean.shape[1])
    elif tuple(mean.shape) == (n, c):
        mean_rows = 0
    else:
        raise ValueError("rcab_tail_bwd: mean must be (n, c) or (n, rows, c)")
    if dsum is not None:
        dsum = _chk(dsum, "dsum")
        if dsum.dim() != 3 or tuple(dsum.shape[::2]) != (n, c):
            raise ValueError("rcab_tail_bwd: dsum must be (n, rows, c)")
        dscale, ds_rows = dsum, int(dsum.shape[1])
    else:
        dscale, ds_rows = plane_sum(d, r), 0
    if grads is None:
        grads = (torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w2), torch.empty_like(b2))
        accumulate = False
    dw1, db1, dw2, db2 = grads
    dr = torch.empty_like(d)
    st = _stream(d)
    _launch("rcab_tail_bwd", 2.0 * d.numel(), 8.0 * d.numel(), d,
            lambda: lib().eavsr_rcab_tail_bwd_f32(_p(d), _p(scale), _p(mean), _p(w1), _p(b1), _p(w2), _p(b2), _p(dscale), _p(dr),
                                                  _p(dw1), _p(db1), _p(dw2), _p(db2), n, c, cr, h * w, mean_rows, ds_rows,
                                                  int(accumulate), st),
            "rcab_tail_bwd")
    return dr, dw1, db1, dw2, db2


def flow_warp_bwd(x: Tensor, flow: Tensor, flow2: Optional[Tensor], dout: Tensor, need_dx: bool, need_dflow: bool):
    x, flow, dout = _chk(x, "x"), _chk(flow, "flow"), _chk(dout, "dout")
    n, c, h, w = x.shape
    if flow2 is not None:
        flow2 = _chk(flow2, "flow2")
    dx = torch.zeros_like(x) if need_dx else None
    dflow = torch.empty_like(flow) if need_dflow else None
    st = _stream(x)
    _launch("flow_warp_bwd", 16.0 * x.numel(), 4.0 * x.numel() * 3, x,
            lambda: lib().eavsr_flow_warp_bwd_f32(_p(x), _p(flow), _p(flow2), _p(dout), _p(dx), _p(dflow), n, c, h, w, st),
            "flow_warp_bwd")
    return dx, dflow


def resize_bilinear_ac_bwd(dout: Tensor, in_shape, scale: float) -> Tensor:
    dout = _chk(dout, "dout")
    n, c, hin, win = in_shape
    hout, wout = dout.shape[2:]
    din = torch.zeros((n, c, hin, win), device=dout.device, dtype=torch.float32)
    st = _stream(dout)
    _launch("resize_bilinear_ac_bwd", 0.0, 4.0 * (dout.numel() + din.numel()), dout,
            lambda: lib().eavsr_resize_bilinear_ac_bwd_f32(_p(dout), _p(din), n, c, hin, win, hout, wout, float(scale), st),
            "resize_bilinear_ac_bwd")
    return din


def pyramid_bwd(dd2: Tensor, dd4: Tensor) -> Tensor:
    dd2, dd4 = _chk(dd2, "dd2"), _chk(dd4, "dd4")
    n, c, h2, w2 = dd2.shape
    din = torch.empty((n, c, 2 * h2, 2 * w2), device=dd2.device, dtype=torch.float32)
    st = _stream(dd2)
    _launch("pyramid_bwd", 0.0, 4.0 * din.numel() * 1.3125, dd2,
            lambda: lib().eavsr_pyramid_bwd_f32(_p(dd2), _p(dd4), _p(din), n * c, 2 * h2, 2 * w2, st), "pyramid_bwd")
    return din


def affine_offsets_bwd(doff: Tensor, dmask: Optional[Tensor], mask: Optional[Tensor], D: int) -> Tensor:
    doff = _chk(doff, "doff")
    n, _, h, w = doff.shape
    with_mask = mask is not None
    if with_mask:
        mask = _chk(mask, "mask")
        dmask = _chk(dmask, "dmask") if dmask is not None else torch.zeros_like(mask)
    dheads = torch.empty((n, (15 if with_mask else 6) * D, h, w), device=doff.device, dtype=torch.float32)
    st = _stream(doff)
    _launch("affine_offsets_bwd", 0.0, 4.0 * (doff.numel() + dheads.numel()), doff,
            lambda: lib().eavsr_affine_offsets_bwd_f32(_p(doff), _p(dmask) if with_mask else None,
                                                       _p(mask) if with_mask else None, _p(dheads), n, D, h, w, st),
            "affine_offsets_bwd")
    return dheads


def conv_wgrad(dy: Tensor, srcs: Sequence[Tensor], ksize: int, out: Optional[Tensor] = None,
               accumulate: bool = False) -> Tensor:
    """Weight gradient (cout, cin_total, k, k) of a conv over the virtual concatenation of `srcs`; written to (or, with
    `accumulate`, added to) `out` when given."""
    dy = _chk(dy, "dy")
    srcs = [_chk(s, f"src{i}") for i, s in enumerate(srcs)]
    n, cout, h, w = dy.shape
    cin = sum(int(s.shape[1]) for s in srcs)
    if out is None:
        dw = torch.empty((cout, cin, ksize, ksize), device=dy.device, dtype=torch.float32)
        accumulate = False
    else:
        dw = out
        if (tuple(dw.shape) != (cout, cin, ksize, ksize) or not dw.is_contiguous() or dw.dtype != torch.float32
                or dw.device != dy.device):
            raise ValueError("conv_wgrad: out must be a contiguous fp32 (cout, cin, k, k) tensor on dy's device")
    acc = int(accumulate)
    blocks = lib().eavsr_conv_wgrad_blocks(n, h, w, ksize)
    if blocks <= 0:
        raise NotImplementedError(f"conv_wgrad: kernel size {ksize}")
    st = _stream(dy)
    if ksize == 1 and max(int(s.shape[1]) for s in srcs) > 64:
        # every 64-channel block of a source in ONE launch + one reduction (DCNv2's 576-channel column tensor: nine of each before)
        ws = torch.empty(blocks * 64 * 64 * max((int(s.shape[1]) + 63) // 64 for s in srcs), device=dy.device, dtype=torch.float32)
        base = 0
        for s in srcs:
            cs = int(s.shape[1])
            for co0 in range(0, cout, 64):
                _launch("conv_wgrad1x1", 2.0 * min(64, cout - co0) * cs * n * h * w, 4.0 * n * h * w * (64 + cs), dy,
                        lambda s=s, cs=cs, co0=co0, base=base: lib().eavsr_conv_wgrad_span_f32(
                            _p(dy), _p(s), _p(dw), _p(ws), n, h, w, cout, co0, cs, cin, base, acc, st), "conv_wgrad_span")
            base += cs
        return dw
    ws = torch.empty(blocks * 64 * 64 * ksize * ksize, device=dy.device, dtype=torch.float32)
    base = 0
    for s in srcs:
        cs = int(s.shape[1])
        for ci0 in range(0, cs, 64):
            for co0 in range(0, cout, 64):
                _launch(f"conv_wgrad{ksize}x{ksize}", 2.0 * min(64, cout - co0) * min(64, cs - ci0) * ksize * ksize * n * h * w,
                        4.0 * n * h * w * 128, dy,
                        lambda s=s, cs=cs, ci0=ci0, co0=co0, base=base: lib().eavsr_conv_wgrad_f32(
                            _p(dy), _p(s), _p(dw), _p(ws), n, h, w, cout, co0, cs, ci0, cin, base + ci0, ksize, acc, st),
                        "conv_wgrad")
        base += cs
    return dw


WGRAD_MAX_SEGMENTS = 8      # csrc/conv_wgrad.hip: WG_MAX_SEG (pointers by value in the kernel arguments)


def _ptr_array(tensors):
    import ctypes as _C
    return (_C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def conv_wgrad_multi(dys: Sequence[Tensor], srcs_list: Sequence[Sequence[Tensor]], ksize: int, out: Tensor,
                     accumulate: bool = False, bias_out: Optional[Tensor] = None) -> Tensor:
    """`conv_wgrad` over several USES of one weight in one launch per (source, 64-channel block): dys[s] / srcs_list[s] are
    the (dY, sources) pairs of use s, all of the same shapes (the frames of the recurrence: eavsrp_model.py:271-324).  The K
    dimension of the weight-gradient GEMM becomes pixels x uses -- a 2 x 96 x 96 training crop has 72 tiles per use for 256
    CUs -- and the per-use accumulation into `out` becomes one slab reduction.  At most WGRAD_MAX_SEGMENTS uses per call.
    bias_out (cout,): the bias gradient sum(dY) is written (or, accumulate=True, added) there as well -- by the bf16x6 3x3 kernel
    itself, which stages dY anyway, by a channel-sum launch otherwise (eavsr_conv_wgrad_bias_multi_f32)."""
    nseg = len(dys)
    if not 1 <= nseg <= WGRAD_MAX_SEGMENTS or len(srcs_list) != nseg:
        raise ValueError(f"conv_wgrad_multi: 1..{WGRAD_MAX_SEGMENTS} segments")
    dys = [_chk(d, "dy") for d in dys]
    srcs_list = [[_chk(s_, "src") for s_ in ss] for ss in srcs_list]
    n, cout, h, w = dys[0].shape
    shapes = [tuple(s_.shape) for s_ in srcs_list[0]]
    for d, ss in zip(dys, srcs_list):
        if tuple(d.shape) != (n, cout, h, w) or [tuple(s_.shape) for s_ in ss] != shapes:
            raise ValueError("conv_wgrad_multi: every segment must have the shapes of the first")
    cin = sum(sh[1] for sh in shapes)
    dw = out
    if (tuple(dw.shape) != (cout, cin, ksize, ksize) or not dw.is_contiguous() or dw.dtype != torch.float32
            or dw.device != dys[0].device):
        raise ValueError("conv_wgrad_multi: out must be a contiguous fp32 (cout, cin, k, k) tensor on dy's device")
    acc = int(accumulate)
    blocks = lib().eavsr_conv_wgrad_blocks(n * nseg, h, w, ksize)
    if blocks <= 0:
        raise NotImplementedError(f"conv_wgrad: kernel size {ksize}")
    if bias_out is not None and (tuple(bias_out.shape) != (cout,) or not bias_out.is_contiguous() or bias_out.dtype != torch.float32
                                 or bias_out.device != dw.device):
        raise ValueError("conv_wgrad_multi: bias_out must be a contiguous fp32 (cout,) tensor on dy's device")
    ws = torch.empty(blocks * (64 * 64 * ksize * ksize + 64), device=dw.device, dtype=torch.float32)
    st = _stream(dw)
    dyl = _ptr_array(dys)
    base = 0
    for si, sh in enumerate(shapes):
        cs = int(sh[1])
        xl = _ptr_array([ss[si] for ss in srcs_list])
        for ci0 in range(0, cs, 64):
            for co0 in range(0, cout, 64):
                db = bias_out if (si == 0 and ci0 == 0) else None      # once per 64-channel block of outputs
                _launch(f"conv_wgrad{ksize}x{ksize}", 2.0 * min(64, cout - co0) * min(64, cs - ci0) * ksize * ksize * n * nseg * h * w,
                        4.0 * n * nseg * h * w * 128, dw,
                        lambda xl=xl, cs=cs, ci0=ci0, co0=co0, base=base, db=db: lib().eavsr_conv_wgrad_bias_multi_f32(
                            dyl, xl, nseg, _p(dw), _p(db), _p(ws), n, h, w, cout, co0, cs, ci0, cin, base + ci0, ksize, acc, st),
                        "conv_wgrad")
        base += cs
    return dw


def channel_sum_multi(tensors: Sequence[Tensor], out: Tensor, accumulate: bool = False) -> Tensor:
    """`channel_sum` over several tensors of one shape in one launch (the bias gradient of conv_wgrad_multi's segments)"""
    nseg = len(tensors)
    if not 1 <= nseg <= WGRAD_MAX_SEGMENTS:
        raise ValueError(f"channel_sum_multi: 1..{WGRAD_MAX_SEGMENTS} segments")
    tensors = [_chk(t, "a") for t in tensors]
    n, c, h, w = tensors[0].shape
    if any(tuple(t.shape) != (n, c, h, w) for t in tensors):
        raise ValueError("channel_sum_multi: every segment must have the shape of the first")
    if tuple(out.shape) != (c,) or not out.is_contiguous() or out.dtype != torch.float32 or out.device != tensors[0].device:
        raise ValueError("channel_sum: out must be a contiguous fp32 (c,) tensor on the input's device")
    if n == 0 or c == 0:
        return out if accumulate else out.zero_()
    st = _stream(out)
    al = _ptr_array(tensors)
    _launch("channel_sum", float(tensors[0].numel() * nseg), 4.0 * tensors[0].numel() * nseg, out,
            lambda: lib().eavsr_channel_sum_multi_f32(al, nseg, _p(out), n, c, h * w, int(accumulate), st), "channel_sum")
    return out


# DCNv2's backward: "sampler" = csrc/dcn_bwd.hip (one kernel, no column tensor; 64 -> 64 channels, 8 groups), "columns" = rounds 1-5
# (im2col -> column tensor -> two GEMM launches -> col2im; any configuration).  EAVSR_DCN_BWD=columns: A/B switch.
DCN_BWD = os.environ.get("EAVSR_DCN_BWD", "sampler")


def dcnv2_bwd_supported(x: Tensor, weight: Tensor, dg: int) -> bool:
    return (DCN_BWD == "sampler" and x.dim() == 4 and int(x.shape[1]) == 64 and tuple(weight.shape) == (64, 64, 3, 3) and dg == 8
            and x.numel() < 2 ** 31)


def dcnv2_bwd(x: Tensor, offset: Tensor, mask: Tensor, weight: Tensor, dy: Tensor, dg: int, need_dx: bool = True,
              dweight: Optional[Tensor] = None, accumulate: bool = False):
    """dx (NCHW or None), doffset, dmask, dweight of modulated_deform_conv2d(x, offset, mask, weight) for the upstream gradient dy, by
    eavsr_dcnv2_bwd_f32 (csrc/dcn_bwd.hip): x goes to the IL8 layout, dx comes back from it.  `dweight` (64, 64, 3, 3): written or,
    accumulate=True, added to (autograd.grad_sink); a new tensor when None."""
    x, offset, mask, dy = _chk(x, "x"), _chk(offset, "offset"), _chk(mask, "mask"), _chk(dy, "dy")
    w_ = _chk(weight.detach(), "weight")
    n, c, h, w = x.shape
    if not dcnv2_bwd_supported(x, weight, dg) or tuple(dy.shape) != (n, 64, h, w):
        raise NotImplementedError("dcnv2_bwd: 64 -> 64 channels, 8 deformable groups, 3x3 (use dcnv2_im2col / dcnv2_col2im)")
    xil = to_il8(x)
    dxil = torch.zeros_like(xil) if need_dx else None
    doff, dmask = torch.empty_like(offset), torch.empty_like(mask)
    if dweight is None:
        dweight, accumulate = torch.empty_like(w_), False
    elif tuple(dweight.shape) != (64, 64, 3, 3) or not dweight.is_contiguous() or dweight.dtype != torch.float32:
        raise ValueError("dcnv2_bwd: dweight must be a contiguous fp32 (64, 64, 3, 3) tensor")
    ws = torch.empty(int(lib().eavsr_dcnv2_bwd_workspace_floats(n, h, w)), device=x.device, dtype=torch.float32)
    st = _stream(x)
    px = float(n) * h * w
    _launch("dcnv2_bwd", 2.0 * 2.0 * 64 * 576 * px, 4.0 * px * (64 * 3 + 27 * dg * 2), x,
            lambda: lib().eavsr_dcnv2_bwd_f32(_p(xil), _p(offset), _p(mask), _p(w_), _p(dy), _p(dxil), _p(doff), _p(dmask), _p(dweight),
                                              _p(ws), n, c, h, w, 64, dg, int(accumulate), st), "dcnv2_bwd")
    dx = None
    if need_dx:
        dx = torch.empty_like(x)
        _launch("il8_to_nchw", 0.0, 8.0 * x.numel(), x,
                lambda: lib().eavsr_il8_to_nchw_f32(_p(dxil), _p(dx), n, c, h, w, st), "il8_to_nchw")
    return dx, doff, dmask, dweight


def dcnv2_im2col(x: Tensor, offset: Tensor, mask: Tensor, dg: int) -> Tensor:
    x, offset, mask = _chk(x, "x"), _chk(offset, "offset"), _chk(mask, "mask")
    n, c, h, w = x.shape
    col = torch.empty((n, c * 9, h, w), device=x.device, dtype=torch.float32)
    st = _stream(x)
    _launch("dcnv2_im2col", 0.0, 4.0 * col.numel(), x,
            lambda: lib().eavsr_dcnv2_im2col_f32(_p(x), _p(offset), _p(mask), _p(col), n, c, h, w, dg, st), "dcnv2_im2col")
    return col


def dcnv2_col2im(x: Tensor, offset: Tensor, mask: Tensor, dcol: Tensor, dg: int, need_dx: bool = True):
    x, offset, mask, dcol = _chk(x, "x"), _chk(offset, "offset"), _chk(mask, "mask"), _chk(dcol, "dcol")
    n, c, h, w = x.shape
    dx = torch.zeros_like(x) if need_dx else None
    doff, dmask = torch.empty_like(offset), torch.empty_like(mask)
    st = _stream(x)
    _launch("dcnv2_col2im", 0.0, 4.0 * dcol.numel(), x,
            lambda: lib().eavsr_dcnv2_col2im_f32(_p(x), _p(offset), _p(mask), _p(dcol), _p(dx), _p(doff), _p(dmask), n, c,
                                                 h, w, dg, st), "dcnv2_col2im")
    return dx, doff, dmask


def gconv3x3(x: Tensor, weight: Tensor, bias: Optional[Tensor], cpg: int, act: Optional[str] = None,
             slope: float = 0.0) -> Tensor:
    x, weight = _chk(x, "x"), _chk(weight.detach(), "weight")
    n, cin, h, w = x.shape
    cout = int(weight.shape[0])
    if cin != cout * cpg or tuple(weight.shape) != (cout, cpg, 3, 3):
        raise ValueError("gconv3x3: one output channel per group, cpg input channels per group")
    b = None if bias is None else _chk(bias.detach(), "bias")
    out = torch.empty((n, cout, h, w), device=x.device, dtype=torch.float32)
    st = _stream(x)
    _launch("gconv3x3", 18.0 * cpg * out.numel(), 4.0 * (x.numel() + out.numel()), x,
            lambda: lib().eavsr_gconv3x3_fwd_f32(_p(x), _p(weight), _p(b), _p(out), n, cout, cpg, h, w, ACT[act],
                                                 float(slope), st), "gconv3x3_fwd")
    return out


def gconv3x3_bwd(g: Tensor, x: Tensor, weight: Tensor, cpg: int, grads=None, accumulate: bool = False):
    """grads=(dweight, dbias) buffers: written -- or, accumulate=True, added to -- in place (autograd.grad_sink)"""
    g, x, weight = _chk(g, "g"), _chk(x, "x"), _chk(weight.detach(), "weight")
    n, cout, h, w = g.shape
    dx = torch.empty_like(x)
    if grads is None:
        dw, db = torch.empty_like(weight), torch.empty(cout, device=g.device, dtype=torch.float32)
        accumulate = False
    else:
        dw, db = grads
        if (tuple(dw.shape) != tuple(weight.shape) or tuple(db.shape) != (cout,) or not dw.is_contiguous() or dw.dtype != torch.float32
                or db.dtype != torch.float32 or dw.device != g.device or db.device != g.device):
            raise ValueError("gconv3x3_bwd: grads = contiguous fp32 (dweight, dbias) of the parameters' shapes on g's device")
    st = _stream(g)
    _launch("gconv3x3_bwd", 36.0 * cpg * g.numel(), 4.0 * (2 * x.numel() + g.numel()), g,
            lambda: lib().eavsr_gconv3x3_bwd_acc_f32(_p(g), _p(x), _p(weight), _p(dx), _p(dw), _p(db), n, cout, cpg, h, w,
                                                     int(bool(accumulate)), st),
            "gconv3x3_bwd")
    return dx, dw, db


# ==========================================================================================
# 16-bit residual backbone (NHWC bf16 / fp16 activations, fp32 accumulate)
# ==========================================================================================
_H16 = {torch.float16: 1, torch.bfloat16: 2, "fp16": 1, "f16": 1, "bf16": 2}
_H16_TORCH = {1: torch.float16, 2: torch.bfloat16}


def h16_code(dtype) -> int:
    try:
        return _H16[dtype]
    except KeyError:
        raise ValueError(f"16-bit dtype expected (torch.float16 / torch.bfloat16 / 'fp16' / 'bf16'), got {dtype!r}")


def _chk_h16(t: Tensor, name: str) -> Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor is on {t.device}; eavsr_amd runs on the GPU only (no CPU path)")
    if t.dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError(f"{name}: dtype {t.dtype}, expected a 16-bit float")
    return t if t.is_contiguous() else t.contiguous()


def to_nhwc_h16(x: Tensor, dtype) -> Tensor:
    """fp32 NCHW (n,c,h,w) -> 16-bit NHWC (n,h,w,c)"""
    x = _chk(x, "x")
    n, c, h, w = x.shape
    code = h16_code(dtype)
    out = torch.empty((n, h, w, c), device=x.device, dtype=_H16_TORCH[code])
    st = _stream(x)
    _launch("nchw_f32_to_nhwc_h16", 0.0, 6.0 * x.numel(), x,
            lambda: lib().eavsr_nchw_f32_to_nhwc_h16(_p(x), _p(out), n, c, h * w, code, st), "nchw_f32_to_nhwc_h16")
    return out


def from_nhwc_h16(x: Tensor, residual: Optional[Tensor] = None) -> Tensor:
    """16-bit NHWC (n,h,w,c) -> fp32 NCHW (n,c,h,w) (+ residual)"""
    x = _chk_h16(x, "x")
    n, h, w, c = x.shape
    code = h16_code(x.dtype)
    out = torch.empty((n, c, h, w), device=x.device, dtype=torch.float32)
    if residual is not None:
        residual = _chk(residual, "residual")
        if residual.shape != out.shape:
            raise ValueError("residual shape mismatch")
    st = _stream(x)
    _launch("nhwc_h16_to_nchw_f32", 0.0, 6.0 * x.numel(), x,
            lambda: lib().eavsr_nhwc_h16_to_nchw_f32(_p(x), _p(residual), _p(out), n, c, h * w, code, st),
            "nhwc_h16_to_nchw_f32")
    return out


_h16_pack_cache = register_weight_cache({})


def _packed_h16(weight: Tensor, code: int) -> Tensor:
    key = (id(weight), weight._version, code)
    hit = _h16_pack_cache.get(key)
    if hit is not None and hit[0]() is weight:
        return hit[1]
    w = _chk(weight.detach(), "weight")
    if tuple(w.shape) != (64, 64, 3, 3):
        raise NotImplementedError("the 16-bit backbone kernel is the 3x3 64->64 convolution")
    packed = torch.empty(64 * 576, device=w.device, dtype=_H16_TORCH[code])
    st = _stream(w)
    with _DeviceOf(w):
        N.check(lib().eavsr_pack_conv3x3_c64_h16(_p(w), _p(packed), code, st), "pack_conv3x3_c64_h16")
    for k in [k for k in _h16_pack_cache if k[0] == id(weight)]:
        _h16_pack_cache.pop(k, None)
    _h16_pack_cache[key] = (weakref.ref(weight, lambda _r, k=key, c=_h16_pack_cache: c.pop(k, None)), packed)
    return packed


def conv3x3_c64_h16(x: Tensor, weight: Tensor, bias: Optional[Tensor], relu: bool = False, chan_partial: bool = False,
                    border: bool = False):
    """x: 16-bit NHWC (n,h,w,64); weight: the fp32 (64,64,3,3) parameter (packed + rounded once per version).
    border=True (with chan_partial): a third result, the `BorderPieces` of the output (for ca_scale_pre_h16(.., border=))."""
    x = _chk_h16(x, "x")
    n, h, w, c = x.shape
    if c != 64:
        raise NotImplementedError("the 16-bit backbone kernel needs 64 channels")
    code = h16_code(x.dtype)
    wp = _packed_h16(weight, code)
    b = None if bias is None else _chk(bias.detach(), "bias")
    out = torch.empty_like(x)
    part = None
    if chan_partial:
        part = torch.empty((n, lib().eavsr_conv_h16_partial_rows(n, h, w), 64), device=x.device, dtype=torch.float32)
    st = _stream(x)
    px = float(n) * h * w
    if border and chan_partial:
        pr, pc = C.c_int32(0), C.c_int32(0)
        N.check(lib().eavsr_conv_h16_border_pieces(h, w, C.byref(pr), C.byref(pc)), "conv_h16_border_pieces")
        stride = max(pr.value, pc.value)
        pieces = BorderPieces(torch.empty((n, 4, stride, 64), device=x.device, dtype=torch.float32), pr.value, pc.value)
        _launch("conv3x3_64to64_h16", 2.0 * 64 * 64 * 9 * px, 2.0 * px * 128, x,
                lambda: lib().eavsr_conv3x3_c64_h16_b(_p(x), _p(wp), _p(b), _p(out), _p(part), _p(pieces.data), stride, n, h, w,
                                                      1 if relu else 0, code, st), "conv3x3_c64_h16_b")
        return out, part, pieces
    _launch("conv3x3_64to64_h16", 2.0 * 64 * 64 * 9 * px, 2.0 * px * 128, x,
            lambda: lib().eavsr_conv3x3_c64_h16(_p(x), _p(wp), _p(b), _p(out), _p(part), n, h, w, 1 if relu else 0, code, st),
            "conv3x3_c64_h16")
    return (out, part) if chan_partial else out




def ca_scale_pre_h16(t: Tensor, partial: Tensor, conv_weight: Tensor, conv_bias: Optional[Tensor], w1: Tensor, b1: Tensor,
                     w2: Tensor, b2: Tensor, border: Optional["BorderPieces"] = None) -> Tensor:
    """The attention of an RCAB (CALayer, networks.py:444-447) BEFORE its second convolution runs: t = the 16-bit NHWC input of
    that convolution, partial = the per-tile channel sums of t from conv3x3_c64_h16(.., relu=True, chan_partial=True),
    conv_weight / conv_bias = the second convolution's parameters.  (n, 64) fp32 for conv3x3_c64_h16(.., skip=, scale=)."""
    t, partial = _chk_h16(t, "t"), _chk(partial, "partial")
    n, h, w, c = t.shape
    if c != 64 or tuple(conv_weight.shape) != (64, 64, 3, 3) or partial.shape[0] != n or partial.shape[2] != 64:
        raise ValueError("ca_scale_pre_h16: 64 channels, a (64, 64, 3, 3) convolution, partial (n, rows, 64)")
    code = h16_code(t.dtype)
    cw = _chk(conv_weight.detach(), "conv_weight")
    cb = None if conv_bias is None else _chk(conv_bias.detach(), "conv_bias")
    w1, b1, w2, b2 = (_chk(v.detach(), "param") for v in (w1, b1, w2, b2))
    cr = int(w1.shape[0])
    scale = torch.empty((n, 64), device=t.device, dtype=torch.float32)
    st = _stream(t)
    if border is not None:      # the border lines as pieces from the first convolution's epilogue: ONE launch
        bd = _chk(border.data, "border pieces")
        if tuple(bd.shape[:2]) != (n, 4) or bd.shape[3] != 64 or bd.shape[2] < max(border.p_rows, border.p_cols):
            raise ValueError("ca_scale_pre_h16: border pieces must be (n, 4, stride >= max(p_rows, p_cols), 64)")
        _launch("ca_scale_pre_h16", 0.0, 4.0 * (partial.numel() + bd.numel()), t,
                lambda: lib().eavsr_ca_scale_pre_pieces(_p(t), _p(partial), int(partial.shape[1]), _p(bd), border.p_rows, border.p_cols,
                                                        int(bd.shape[2]), _p(cw), _p(cb), _p(w1), _p(b1), _p(w2), _p(b2), _p(scale),
                                                        n, h, w, cr, code, st), "ca_scale_pre_pieces")
        return scale
    ws = torch.empty(int(lib().eavsr_ca_scale_pre_ws_floats(n)), device=t.device, dtype=torch.float32)
    _launch("ca_scale_pre_h16", 0.0, 4.0 * partial.numel(), t,
            lambda: lib().eavsr_ca_scale_pre_h16(_p(t), _p(partial), int(partial.shape[1]), _p(cw), _p(cb), _p(w1), _p(b1), _p(w2), _p(b2),
                                                 _p(scale), _p(ws), n, h, w, cr, code, st), "ca_scale_pre_h16")
    return scale


def ca_scale_pre(t: Tensor, partial: Tensor, conv_weight: Tensor, conv_bias: Optional[Tensor], w1: Tensor, b1: Tensor,
                 w2: Tensor, b2: Tensor, border: Optional["BorderPieces"] = None) -> Tensor:
    """ca_scale_pre_h16 for the fp32 NCHW path: t (n, 64, h, w) = ReLU(conv1(x)), partial = its per-tile channel sums (the first
    convolution's chan_partial), conv_weight / conv_bias = the SECOND convolution's.  (n, 64) for conv2d(.., residual=, res_scale=)."""
    t, partial = _chk(t, "t"), _chk(partial, "partial")
    n, c, h, w = t.shape
    if c != 64 or tuple(conv_weight.shape) != (64, 64, 3, 3) or partial.shape[0] != n or partial.shape[2] != 64:
        raise ValueError("ca_scale_pre: 64 channels, a (64, 64, 3, 3) convolution, partial (n, tiles, 64)")
    cw = _chk(conv_weight.detach(), "conv_weight")
    cb = None if conv_bias is None else _chk(conv_bias.detach(), "conv_bias")
    w1, b1, w2, b2 = (_chk(v.detach(), "param") for v in (w1, b1, w2, b2))
    cr = int(w1.shape[0])
    scale = torch.empty((n, 64), device=t.device, dtype=torch.float32)
    st = _stream(t)
    if border is not None:      # the border lines as pieces from the first convolution's epilogue: ONE launch
        bd = _chk(border.data, "border pieces")
        if tuple(bd.shape[:2]) != (n, 4) or bd.shape[3] != 64 or bd.shape[2] < max(border.p_rows, border.p_cols):
            raise ValueError("ca_scale_pre: border pieces must be (n, 4, stride >= max(p_rows, p_cols), 64)")
        _launch("ca_scale_pre", 0.0, 4.0 * (partial.numel() + bd.numel()), t,
                lambda: lib().eavsr_ca_scale_pre_pieces(_p(t), _p(partial), int(partial.shape[1]), _p(bd), border.p_rows, border.p_cols,
                                                        int(bd.shape[2]), _p(cw), _p(cb), _p(w1), _p(b1), _p(w2), _p(b2), _p(scale),
                                                        n, h, w, cr, 0, st), "ca_scale_pre_pieces")
        return scale
    ws = torch.empty(int(lib().eavsr_ca_scale_pre_ws_floats(n)), device=t.device, dtype=torch.float32)
    _launch("ca_scale_pre", 0.0, 4.0 * partial.numel(), t,
            lambda: lib().eavsr_ca_scale_pre_f32(_p(t), _p(partial), int(partial.shape[1]), _p(cw), _p(cb), _p(w1), _p(b1), _p(w2), _p(b2),
                                                 _p(scale), _p(ws), n, h, w, cr, st), "ca_scale_pre")
    return scale


def conv3x3_c64_h16_res(x: Tensor, weight: Tensor, bias: Optional[Tensor], skip: Tensor, scale: Tensor) -> Tensor:
    """skip + scale[n, co] * (conv3x3(x) + bias), rounded once: the RCAB's second convolution with its tail as the epilogue
    (networks.py:461-464); x, skip 16-bit NHWC (n,h,w,64), scale (n, 64) fp32 from ca_scale_pre_h16"""
    x, skip, scale = _chk_h16(x, "x"), _chk_h16(skip, "skip"), _chk(scale, "scale")
    n, h, w, c = x.shape
    if c != 64 or skip.shape != x.shape or skip.dtype != x.dtype or tuple(scale.shape) != (n, 64):
        raise ValueError("conv3x3_c64_h16_res: x / skip (n,h,w,64) of one 16-bit type, scale (n, 64)")
    code = h16_code(x.dtype)
    wp = _packed_h16(weight, code)
    b = None if bias is None else _chk(bias.detach(), "bias")
    out = torch.empty_like(x)
    st = _stream(x)
    px = float(n) * h * w
    _launch("conv3x3_64to64_h16", 2.0 * 64 * 64 * 9 * px, 3.0 * px * 128, x,
            lambda: lib().eavsr_conv3x3_c64_h16_res(_p(x), _p(wp), _p(b), _p(out), _p(skip), _p(scale), n, h, w, code, st),
            "conv3x3_c64_h16_res")
    return out


# The one-launch RCAB convolutions stream their weights per tile (11.5 us per tile whatever the launch); the resident-weights
# kernel amortises its weight load and prologue over a workgroup's tiles (7.1 us per tile and convolution at 4 tiles per
# workgroup, 5 us at 8).  Measured (tools/gpu_rcab_h16_time.py): 4 x 256 x 256 (4 tiles per workgroup) 49.7 us against 52.3 for the two
# launches, 2 x 180 x 320 (1.8) 25.3 against 31.0, 1 x 540 x 960 (8) 94.1 against 81.7 -- so the fused form is the default up to
# RCAB_FUSED_MAX_TILES tiles per launch (EAVSR_RCAB_FUSED_MAX_TILES overrides).
RCAB_FUSED_MAX_TILES = int(os.environ.get("EAVSR_RCAB_FUSED_MAX_TILES", "1536"))


def rcab_convs_h16_preferred(x: Tensor) -> bool:
    n, h, w, _ = x.shape
    return n * ((h + 7) // 8) * ((w + 31) // 32) <= RCAB_FUSED_MAX_TILES


def rcab_convs_h16(x: Tensor, w1: Tensor, b1: Optional[Tensor], w2: Tensor, b2: Optional[Tensor], chan_partial: bool = True):
    """r = conv3x3(ReLU(conv3x3(x))) of one RCAB (networks.py:461-462) as ONE launch on 16-bit NHWC tensors
    (csrc/rcab_h16.hip: streamed weights, the intermediate lives in LDS); with `chan_partial` also the channel sums of r for the
    channel attention, in the row layout of conv3x3_c64_h16."""
    require_lab("ops.rcab_convs_h16 (conv -> ReLU -> conv of a 16-bit RCAB as one launch)")
    x = _chk_h16(x, "x")
    n, h, w, c = x.shape
    if c != 64:
        raise NotImplementedError("the 16-bit backbone kernel needs 64 channels")
    code = h16_code(x.dtype)
    wp1, wp2 = _packed_h16(w1, code), _packed_h16(w2, code)
    bb1 = None if b1 is None else _chk(b1.detach(), "bias")
    bb2 = None if b2 is None else _chk(b2.detach(), "bias")
    out = torch.empty_like(x)
    part = None
    if chan_partial:
        part = torch.empty((n, lib().eavsr_rcab_h16_partial_rows(n, h, w), 64), device=x.device, dtype=torch.float32)
    st = _stream(x)
    px = float(n) * h * w
    _launch("rcab_convs_h16", 2.0 * 2.0 * 64 * 64 * 9 * px, 2.0 * px * 128, x,
            lambda: lib().eavsr_rcab_convs_h16(_p(x), _p(wp1), _p(bb1), _p(wp2), _p(bb2), _p(out), _p(part), n, h, w, code, st),
            "rcab_convs_h16")
    return (out, part) if chan_partial else out


_h16_ps_cache = register_weight_cache({})
_h16_last_cache = register_weight_cache({})


def _packed_h16_ps2(weight: Tensor, bias: Optional[Tensor], code: int):
    """(256, 64, 3, 3) weight of a conv + PixelShuffle(2) stage -> four packed 64 -> 64 matrices (slice k = 2 dy + dx: the output
    channels 4 c + k as channel c) and the bias in the same order (4 x 64)"""
    key = (id(weight), weight._version, None if bias is None else (id(bias), bias._version), code)
    hit = _h16_ps_cache.get(key)
    if hit is not None and hit[0]() is weight:
        return hit[1], hit[2]
    w = _chk(weight.detach(), "weight")
    if tuple(w.shape) != (256, 64, 3, 3):
        raise NotImplementedError("the 16-bit pixel-shuffle stage is the 3x3 64 -> 256 convolution")
    sl = w.view(64, 4, 64, 3, 3).permute(1, 0, 2, 3, 4).contiguous()      # [k][c][ci][3][3]
    packed = torch.empty(4, 64 * 576, device=w.device, dtype=_H16_TORCH[code])
    st = _stream(w)
    with _DeviceOf(w):
        for k in range(4):
            N.check(lib().eavsr_pack_conv3x3_c64_h16(_p(sl[k]), _p(packed[k]), code, st), "pack_conv3x3_c64_h16")
    b4 = None if bias is None else _chk(bias.detach(), "bias").view(64, 4).t().contiguous()
    for k in [k for k in _h16_ps_cache if k[0] == id(weight)]:
        _h16_ps_cache.pop(k, None)
    _h16_ps_cache[key] = (weakref.ref(weight, lambda _r, k=key, c=_h16_ps_cache: c.pop(k, None)), packed, b4)
    return packed, b4


def conv3x3_c64_h16_act(x: Tensor, weight: Tensor, bias: Optional[Tensor], act: Optional[str] = None, slope: float = 0.0,
                        pixel_shuffle2: bool = False) -> Tensor:
    """The upsampling tail in the 16-bit modes: x 16-bit NHWC (n,h,w,64); weight (64,64,3,3) -> act(conv + bias), or, with
    pixel_shuffle2, weight (256,64,3,3) -> act(PixelShuffle(2)(conv + bias)) as (n,2h,2w,64) (eavsrp_model.py:343-357)"""
    x = _chk_h16(x, "x")
    n, h, w, c = x.shape
    if c != 64:
        raise NotImplementedError("the 16-bit backbone kernel needs 64 channels")
    code = h16_code(x.dtype)
    if pixel_shuffle2:
        wp, b = _packed_h16_ps2(weight, bias, code)
        out = torch.empty((n, 2 * h, 2 * w, 64), device=x.device, dtype=x.dtype)
    else:
        wp = _packed_h16(weight, code)
        b = None if bias is None else _chk(bias.detach(), "bias")
        out = torch.empty_like(x)
    a = ACT[act]
    st = _stream(x)
    px = float(n) * h * w
    co = 256 if pixel_shuffle2 else 64
    _launch("conv3x3_64to256_h16_ps2" if pixel_shuffle2 else "conv3x3_64to64_h16", 2.0 * 64 * co * 9 * px, px * (128 + 2 * co), x,
            lambda: lib().eavsr_conv3x3_c64_h16_act(_p(x), _p(wp), _p(b), _p(out), n, h, w, a, float(slope), 1 if pixel_shuffle2 else 0, code, st),
            "conv3x3_c64_h16_act")
    return out


def conv3x3_c64to3_h16(x: Tensor, weight: Tensor, bias: Optional[Tensor], residual: Optional[Tensor] = None) -> Tensor:
    """conv_last of the tail in the 16-bit modes: x 16-bit NHWC (n,h,w,64), weight (3,64,3,3) fp32 -> fp32 NCHW (n,3,h,w) (+ residual)"""
    x = _chk_h16(x, "x")
    n, h, w, c = x.shape
    if c != 64 or tuple(weight.shape) != (3, 64, 3, 3):
        raise NotImplementedError("the 16-bit conv_last kernel is the 3x3 64 -> 3 convolution")
    key = (id(weight), weight._version, x.dtype)
    hit = _h16_last_cache.get(key)
    if hit is not None and hit[0]() is weight:
        wt = hit[1]
    else:      # [ky][kx][8-channel block][co][8 channels], rounded to the activation type (scalar loads in the kernel)
        wt = _chk(weight.detach(), "weight").view(3, 8, 8, 3, 3).permute(3, 4, 1, 0, 2).contiguous().to(x.dtype)
        for k in [k for k in _h16_last_cache if k[0] == id(weight)]:
            _h16_last_cache.pop(k, None)
        _h16_last_cache[key] = (weakref.ref(weight, lambda _r, k=key, c=_h16_last_cache: c.pop(k, None)), wt)
    b = None if bias is None else _chk(bias.detach(), "bias")
    r = None if residual is None else _chk(residual, "residual")
    if r is not None and tuple(r.shape) != (n, 3, h, w):
        raise ValueError("residual must be (n, 3, h, w)")
    out = torch.empty((n, 3, h, w), device=x.device, dtype=torch.float32)
    st = _stream(x)
    px = float(n) * h * w
    _launch("conv3x3_64to3_h16", 2.0 * 64 * 3 * 9 * px, px * (128 + 12 + (12 if r is not None else 0)), x,
            lambda: lib().eavsr_conv3x3_c64to3_h16(_p(x), _p(wt), _p(b), _p(r), _p(out), n, h, w, h16_code(x.dtype), st),
            "conv3x3_c64to3_h16")
    return out


_h16_pack5_cache = register_weight_cache({})


def _packed5_h16(wcat: Tensor, code: int) -> Tensor:
    """(cout, 64, 5, 5) fp32 (a parameter or the cached concatenation of several) -> the 16-bit MFMA-fragment order"""
    key = (id(wcat), wcat._version, code)
    hit = _h16_pack5_cache.get(key)
    if hit is not None and hit[0]() is wcat:
        return hit[1]
    if wcat.dim() != 4 or tuple(wcat.shape[1:]) != (64, 5, 5) or wcat.shape[0] > 128:
        raise NotImplementedError("the 16-bit heads kernel is the 5x5 convolution 64 -> (<= 128) channels")
    packed = torch.empty(int(lib().eavsr_conv5x5_c64_h16_weight_bytes()) // 2, device=wcat.device, dtype=_H16_TORCH[code])
    st = _stream(wcat)
    with _DeviceOf(wcat):
        N.check(lib().eavsr_pack_conv5x5_c64_h16(_p(wcat), _p(packed), int(wcat.shape[0]), code, st), "pack_conv5x5_c64_h16")
    for k in [k for k in _h16_pack5_cache if k[0] == id(wcat)]:
        _h16_pack5_cache.pop(k, None)
    _h16_pack5_cache[key] = (weakref.ref(wcat, lambda _r, k=key, c=_h16_pack5_cache: c.pop(k, None)), packed)
    return packed


def conv5x5_c64_h16(x: Tensor, weights, biases) -> Tensor:
    """the predictor's 5x5 heads in the 16-bit modes: x 16-bit NHWC (n,h,w,64); weights / biases: one fp32 (cout,64,5,5) /
    (cout,) parameter or a list of them (concatenated along the output channels); returns fp32 NCHW (n, cout, h, w)"""
    x = _chk_h16(x, "x")
    n, h, w, c = x.shape
    if c != 64:
        raise NotImplementedError("the 16-bit heads kernel needs 64 input channels")
    code = h16_code(x.dtype)
    weights = list(weights) if isinstance(weights, (list, tuple)) else [weights]
    biases = list(biases) if isinstance(biases, (list, tuple)) else [biases]
    wcat = _cat_weights(weights)
    wp = _packed5_h16(wcat, code)
    cout = int(wcat.shape[0])
    b = None
    if any(bb is not None for bb in biases):
        if not all(bb is not None for bb in biases):
            raise ValueError("conv5x5_c64_h16: all heads need a bias or none")
        b = _cat_weights(biases)
    out = torch.empty((n, cout, h, w), device=x.device, dtype=torch.float32)
    st = _stream(x)
    px = float(n) * h * w
    _launch(f"conv5x5_64to{cout}_h16", 2.0 * 64 * cout * 25 * px, px * (128 + 4.0 * cout), x,
            lambda: lib().eavsr_conv5x5_c64_h16(_p(x), _p(wp), _p(b), _p(out), n, h, w, cout, code, st), "conv5x5_c64_h16")
    return out


def scale_residual_h16(r: Tensor, scale: Tensor, x: Tensor) -> Tensor:
    r, x, scale = _chk_h16(r, "r"), _chk_h16(x, "x"), _chk(scale, "scale")
    n, h, w, c = r.shape
    if x.shape != r.shape or x.dtype != r.dtype or tuple(scale.shape) != (n, c):
        raise ValueError("scale_residual_h16: shape / dtype mismatch")
    out = torch.empty_like(r)
    code = h16_code(r.dtype)
    st = _stream(r)
    _launch("scale_residual_h16", 2.0 * r.numel(), 6.0 * r.numel(), r,
            lambda: lib().eavsr_scale_residual_h16(_p(r), _p(scale), _p(x), _p(out), n, c, h * w, code, st),
            "scale_residual_h16")
    return out
