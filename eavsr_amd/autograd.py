"""Differentiable front of eavsr_amd.ops (training step).

Every function here has the signature of its namesake in ops.py.  When no input needs a gradient
(inference, `torch.no_grad()`), it forwards straight to the fused forward kernel.  Otherwise it
goes through a torch.autograd.Function whose backward launches the HIP backward kernels
(csrc/backward.hip, backward_dcn.hip, conv_wgrad.hip; dgrad reuses the MFMA conv kernel with the
transposed, flipped weight).  PyTorch only records the graph -- no gradient is computed by an
ATen kernel on the hot path.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple, Union

import os
import weakref

import torch
from torch.autograd import Function

from . import ops

Tensor = torch.Tensor


def _needs_grad(*ts) -> bool:
    if not torch.is_grad_enabled():
        return False
    for t in ts:
        if isinstance(t, (list, tuple)):
            if _needs_grad(*t):
                return True
        elif isinstance(t, torch.Tensor) and t.requires_grad:
            return True
    return False


# ------------------------------------------------------------------------------------------ gradient sink
class grad_sink:
    """Context manager for `loss.backward()` of the training step.

    A convolution's parameters are used once per frame and branch (7-28 times per step).  Autograd sums the
    per-use gradients with one ATen add per use and tensor (~6,000 launches of a few microseconds per step).
    Inside this context the wgrad / bias kernels add every use into one buffer per call site instead
    (`accumulate` of eavsr_conv_wgrad_f32 / eavsr_channel_sum_f32, fixed order: the order of backward) and the
    Function returns no gradient for the parameter; `flush()` (called on exit) hands the buffers to `.grad`.
    Only leaf parameters take part; everything else goes through autograd as usual.  Gradient hooks of the sunk
    parameters do not fire (shard.GradientAllReducer.finish() launches their buckets after the flush)."""

    _active: Optional["grad_sink"] = None

    def __init__(self):
        self.entries = {}

    def __enter__(self):
        if grad_sink._active is not None:
            raise RuntimeError("grad_sink: already active")
        grad_sink._active = self
        return self

    def __exit__(self, et, ev, tb):
        grad_sink._active = None
        if et is None:
            self.flush()
        self.entries = {}
        _chain.clear()      # (what consecutive RCABs handed each other during this forward / backward: see _Chain)
        return False

    @staticmethod
    def eligible(params) -> bool:
        return grad_sink._active is not None and all(
            isinstance(p, torch.Tensor) and p.is_leaf and p.requires_grad and p.is_cuda for p in params)

    # uses of one weight that wait for their (batched) weight-gradient launch: at most this many per launch
    BATCH = ops.WGRAD_MAX_SEGMENTS
    # uses per call site in the LAST backward that ended: a call site whose expected uses are all in launches at once instead of
    # holding its dY and saved sources until flush() -- most convolutions outside the recurrence are used once per step, and
    # their activations would otherwise stay alive until backward ends (peak memory, baked into the graph's pool)
    _last_uses = {}

    def add_use(self, ws, bs, k, g, srcs):
        """One use of the weights `ws` (+ biases `bs`): dY = g over the sources `srcs`.  The gradient launch is DEFERRED: uses of
        the same call site and shapes wait until BATCH of them are there (or backward ends) and go through
        ops.conv_wgrad_multi / channel_sum_multi as segments of ONE launch -- a weight of the recurrent path is used once per
        frame, and a 2 x 96 x 96 crop fills 72 tiles into 256 CUs per use.  Same sums, K extended over the uses; fixed order."""
        key = tuple(id(w) for w in ws)
        e = self.entries.get(key)
        if e is None:
            cout = sum(int(w.shape[0]) for w in ws)
            dW = torch.empty((cout, int(ws[0].shape[1]), k, k), device=ws[0].device, dtype=torch.float32)
            db = torch.empty((cout,), device=ws[0].device, dtype=torch.float32) if bs is not None else None
            e = {"ws": list(ws), "bs": None if bs is None else list(bs), "dW": dW, "db": db, "k": k, "written": False, "pending": [],
                 "uses": 0, "key": key}
            self.entries[key] = e
        if e["pending"] and (tuple(e["pending"][0][0].shape) != tuple(g.shape)
                             or [tuple(s_.shape) for s_ in e["pending"][0][1]] != [tuple(s_.shape) for s_ in srcs]):
            self._launch(e)       # another shape at the same call site: what waits goes first
        e["pending"].append((g, list(srcs)))
        e["uses"] += 1
        if len(e["pending"]) >= self.BATCH or e["uses"] == grad_sink._last_uses.get(key, -1):
            self._launch(e)       # a full batch, or the last use this call site had in the previous backward

    @staticmethod
    def _launch(e):
        pend = e["pending"]
        if not pend:
            return
        acc = e["written"]
        # (the bias gradient rides in the weight-gradient launch: the 3x3 kernel stages dY anyway)
        ops.conv_wgrad_multi([g for g, _ in pend], [ss for _, ss in pend], e["k"], out=e["dW"], accumulate=acc, bias_out=e["db"])
        e["written"] = True
        e["pending"] = []

    def raw(self, params):
        """Buffers of the parameters' shapes that a backward kernel writes (first use) or adds to (later uses) itself:
        (buffers, accumulate?).  For gradients that are not a convolution's (the channel-attention MLP of the RCAB tail)."""
        key = ("raw",) + tuple(id(p) for p in params)
        e = self.entries.get(key)
        if e is None:
            e = {"raw": [torch.empty_like(p) for p in params], "params": list(params), "written": False}
            self.entries[key] = e
        acc = e["written"]
        e["written"] = True
        return e["raw"], acc

    def flush(self):
        for e in self.entries.values():
            if "raw" in e:
                for p, g in zip(e["params"], e["raw"]):
                    p.grad = g if p.grad is None else p.grad + g
                continue
            self._launch(e)
            grad_sink._last_uses[e["key"]] = e["uses"]
            ws, bs, dW, db = e["ws"], e["bs"], e["dW"], e["db"]
            c0 = 0
            for i, w_ in enumerate(ws):
                c = int(w_.shape[0])
                for p, g in ((w_, dW[c0:c0 + c]), (None if bs is None else bs[i], None if db is None else db[c0:c0 + c])):
                    if p is not None:
                        p.grad = g if p.grad is None else p.grad + g
                c0 += c
        self.entries = {}


# ------------------------------------------------------------------------------------------ conv
# dgrad operand of a conv: the weight with its taps flipped and (cout, cin) transposed, for the source channels
# c0 .. c0+cs.  A weight is used once per frame and branch, so within a step the same operand is asked for 7-28
# times: it is built once per weight version (and, being the same tensor object, its packed form is then also a
# hit in ops.pack_cache).
_dgrad_cache = ops.register_weight_cache({})


def _dgrad_weight(ws, c0: int, cs: int) -> Tensor:
    key = tuple((id(w), w._version) for w in ws) + (c0, cs)
    hit = _dgrad_cache.get(key)
    if hit is not None and all(r() is w for r, w in zip(hit[0], ws)):
        return hit[1]
    W = ws[0] if len(ws) == 1 else torch.cat(ws, 0)
    wt = W.detach()[:, c0:c0 + cs].flip(2, 3).transpose(0, 1).contiguous()   # (cs, cout, k, k)
    ids = {id(w) for w in ws}
    for k_ in [k_ for k_ in _dgrad_cache if k_[-2:] == (c0, cs) and any(i in ids for i, _ in k_[:-2])]:
        _dgrad_cache.pop(k_, None)
    refs = tuple(weakref.ref(w, lambda _r, k_=key, c=_dgrad_cache: c.pop(k_, None)) for w in ws)
    _dgrad_cache[key] = (refs, wt)
    return wt


class _ConvFn(Function):
    @staticmethod
    def forward(ctx, act, slope, n_w, has_bias, has_res, n_src, *ts):
        ws = list(ts[:n_w])
        bs = list(ts[n_w:2 * n_w]) if has_bias else [None] * n_w
        o = 2 * n_w if has_bias else n_w
        res = ts[o] if has_res else None
        o += 1 if has_res else 0
        srcs = list(ts[o:o + n_src])
        if act is not None and has_res:
            raise NotImplementedError("conv2d backward: activation together with a residual is not used on the path")
        out = ops.conv2d(srcs, ws, bs if has_bias else None, act=act, slope=slope, residual=res)
        ctx.meta = (act, slope, n_w, has_bias, has_res, n_src)
        ctx.params = (ws, bs if has_bias else None)    # the caller's tensor objects (grad_sink keys on them)
        ctx.save_for_backward(*ws, *srcs, *( [out] if act is not None else []))
        return out

    @staticmethod
    def backward(ctx, dout):
        act, slope, n_w, has_bias, has_res, n_src = ctx.meta
        saved = ctx.saved_tensors
        ws, srcs = list(saved[:n_w]), list(saved[n_w:n_w + n_src])
        dout = dout.contiguous()
        g = ops.act_bwd(dout, saved[-1], act, slope) if act is not None else dout
        k = int(ws[0].shape[-1])
        need = ctx.needs_input_grad[6:]
        need_w = need[:n_w]
        o = 2 * n_w if has_bias else n_w
        need_res = need[o] if has_res else False
        o += 1 if has_res else 0
        need_src = need[o:o + n_src]
        dws: List[Optional[Tensor]] = [None] * n_w
        pw, pb = ctx.params
        if all(need_w) and (not has_bias or all(need[n_w:2 * n_w])) and grad_sink.eligible(pw + (pb or [])):
            grad_sink._active.add_use(pw, pb if has_bias else None, k, g, srcs)      # launched in batches of uses (grad_sink)
            need_w = [False] * n_w
            has_bias_grad = False
        else:
            has_bias_grad = has_bias
        if any(need_w):
            dW = ops.conv_wgrad(g, srcs, k)
            c0 = 0
            for i, w_ in enumerate(ws):
                dws[i] = dW[c0:c0 + w_.shape[0]] if need_w[i] else None
                c0 += w_.shape[0]
        dbs: List[Optional[Tensor]] = []
        if has_bias_grad:
            db = ops.channel_sum(g)
            c0 = 0
            for w_ in ws:
                dbs.append(db[c0:c0 + w_.shape[0]])
                c0 += w_.shape[0]
        elif has_bias:
            dbs = [None] * n_w
        dsrcs: List[Optional[Tensor]] = []
        c0 = 0
        for i, s in enumerate(srcs):
            cs = int(s.shape[1])
            if need_src[i]:
                dsrcs.append(ops.conv2d(g, _dgrad_weight(ws, c0, cs), None))
            else:
                dsrcs.append(None)
            c0 += cs
        return (None,) * 6 + tuple(dws) + tuple(dbs) + ((dout if need_res else None,) if has_res else ()) + tuple(dsrcs)


def conv2d(srcs, weight, bias=None, act=None, slope=0.0, residual=None, chan_partial=False, ca=None, ca_out=False,
           pixel_shuffle2=False):
    if isinstance(srcs, torch.Tensor):
        srcs = [srcs]
    ws = [weight] if isinstance(weight, torch.Tensor) else list(weight)
    bs = [bias] if (bias is None or isinstance(bias, torch.Tensor)) else list(bias)
    if not _needs_grad(srcs, ws, bs, residual):
        return ops.conv2d(srcs, weight, bias, act=act, slope=slope, residual=residual, chan_partial=chan_partial,
                          ca=ca, ca_out=ca_out, pixel_shuffle2=pixel_shuffle2)
    if pixel_shuffle2:      # training: the plain conv (with its backward kernels), shuffled by torch
        return torch.nn.functional.pixel_shuffle(conv2d(srcs, weight, bias, act=act, slope=slope), 2)
    if chan_partial or ca is not None:
        raise NotImplementedError("the fused channel-attention paths are inference-only; training uses rcab_tail")
    has_bias = bs[0] is not None
    ts = list(ws) + (list(bs) if has_bias else []) + ([residual] if residual is not None else []) + list(srcs)
    return _ConvFn.apply(act, float(slope), len(ws), has_bias, residual is not None, len(srcs), *ts)


# ------------------------------------------------------------------------------------------ flow_warp
class _FlowWarpFn(Function):
    @staticmethod
    def forward(ctx, x, flow, flow2):
        ctx.save_for_backward(x, flow, *( [flow2] if flow2 is not None else []))
        return ops.flow_warp(x, flow, "zeros", flow2=flow2)

    @staticmethod
    def backward(ctx, dout):
        saved = ctx.saved_tensors
        x, flow = saved[0], saved[1]
        flow2 = saved[2] if len(saved) > 2 else None
        need_dx = ctx.needs_input_grad[0]
        need_df = ctx.needs_input_grad[1] or (flow2 is not None and ctx.needs_input_grad[2])
        dx, dflow = ops.flow_warp_bwd(x, flow, flow2, dout.contiguous(), need_dx, need_df)
        return dx, (dflow if ctx.needs_input_grad[1] else None), (dflow if flow2 is not None and ctx.needs_input_grad[2] else None)


def flow_warp(x, flow, padding_mode="zeros", flow2=None, flow_layout="nchw", interpolation="bilinear",
              align_corners=True):
    if not _needs_grad(x, flow, flow2):
        return ops.flow_warp(x, flow, padding_mode=padding_mode, flow2=flow2, flow_layout=flow_layout,
                             interpolation=interpolation, align_corners=align_corners)
    if padding_mode != "zeros" or interpolation != "bilinear" or not align_corners:
        raise NotImplementedError("flow_warp backward exists for the configuration the trainable path uses (bilinear, "
                                  "padding_mode='zeros', align_corners=True; border is only used inside the frozen SPyNet)")
    if flow_layout == "nhwc":
        flow = flow.permute(0, 3, 1, 2)
        flow2 = None if flow2 is None else flow2.permute(0, 3, 1, 2)
    return _FlowWarpFn.apply(x, flow.contiguous(), None if flow2 is None else flow2.contiguous())


# ------------------------------------------------------------------------------------------ DCNv2
_dcn_wt_cache = ops.register_weight_cache({})


def _dcn_wt(weight: Tensor) -> Tensor:
    """(cin * 9, cout, 1, 1): the DCNv2 weight as the 1x1 convolution that maps dOut to the column gradients; once per weight version"""
    key = (id(weight), weight._version)
    hit = _dcn_wt_cache.get(key)
    if hit is not None and hit[0]() is weight:
        return hit[1]
    cout, cin = int(weight.shape[0]), int(weight.shape[1])
    wt = weight.detach().reshape(cout, cin * 9).t().contiguous().view(cin * 9, cout, 1, 1)
    for k_ in [k_ for k_ in _dcn_wt_cache if k_[0] == id(weight)]:
        _dcn_wt_cache.pop(k_, None)
    _dcn_wt_cache[key] = (weakref.ref(weight, lambda _r, k_=key, c=_dcn_wt_cache: c.pop(k_, None)), wt)
    return wt


class _DcnFn(Function):
    @staticmethod
    def forward(ctx, x, offset, mask, weight, bias, dg):
        ctx.dg = dg
        ctx.has_bias = bias is not None
        ctx.params = [weight] + ([bias] if bias is not None else [])      # the caller's tensor objects (grad_sink keys on them)
        ctx.save_for_backward(x, offset, mask, weight)
        return ops.modulated_deform_conv2d(x, offset, mask, weight, bias, 1, 1, 1, 1, dg)

    @staticmethod
    def backward(ctx, dout):
        x, offset, mask, weight = ctx.saved_tensors
        dg = ctx.dg
        dout = dout.contiguous()
        cout, cin = weight.shape[0], weight.shape[1]
        need_w, need_b = ctx.needs_input_grad[3], ctx.has_bias and ctx.needs_input_grad[4]
        if ops.dcnv2_bwd_supported(x, weight, dg):
            # round 6: the whole backward on the sampler's side (csrc/dcn_bwd.hip): no column tensor, no GEMM launches, no col2im
            if need_w and (need_b or not ctx.has_bias) and grad_sink.eligible(ctx.params):
                bufs, acc = grad_sink._active.raw(ctx.params)      # the uses of the weight across the frames add into one buffer
                dx, doff, dmask, _ = ops.dcnv2_bwd(x, offset, mask, weight, dout, dg, need_dx=ctx.needs_input_grad[0], dweight=bufs[0],
                                                   accumulate=acc)
                if ctx.has_bias:
                    ops.channel_sum(dout, out=bufs[1], accumulate=acc)
                return dx, doff, dmask, None, None, None
            dx, doff, dmask, dW = ops.dcnv2_bwd(x, offset, mask, weight, dout, dg, need_dx=ctx.needs_input_grad[0])
            return dx, doff, dmask, (dW if need_w else None), (ops.channel_sum(dout) if need_b else None), None
        col = ops.dcnv2_im2col(x, offset, mask, dg)                              # (n, cin*9, h, w)
        dW = db = None
        if need_w and (need_b or not ctx.has_bias) and grad_sink.eligible(ctx.params):
            # the uses of the weight across the frames add into one buffer (was: one ATen add per use and tensor)
            bufs, acc = grad_sink._active.raw(ctx.params)
            ops.conv_wgrad(dout, [col], 1, out=bufs[0].view(cout, cin * 9, 1, 1), accumulate=acc)
            if ctx.has_bias:
                ops.channel_sum(dout, out=bufs[1], accumulate=acc)
        else:
            dW = ops.conv_wgrad(dout, [col], 1).view(cout, cin, 3, 3) if need_w else None
            db = ops.channel_sum(dout) if need_b else None
        dcol = ops.conv2d(dout, _dcn_wt(weight), None)                           # W^T . dOut
        dx, doff, dmask = ops.dcnv2_col2im(x, offset, mask, dcol, dg, need_dx=ctx.needs_input_grad[0])
        return dx, doff, dmask, dW, db, None


def modulated_deform_conv2d(input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                            deform_groups=1):
    if not _needs_grad(input, offset, mask, weight, bias):
        return ops.modulated_deform_conv2d(input, offset, mask, weight, bias, stride, padding, dilation, groups,
                                           deform_groups)
    one = lambda v: int(v[0]) if isinstance(v, (tuple, list)) else int(v)
    if tuple(weight.shape[2:]) != (3, 3) or one(stride) != 1 or one(padding) != 1 or one(dilation) != 1 or groups != 1:
        raise NotImplementedError("modulated_deform_conv2d: only the reference configuration (3x3, 1, 1, 1, 1)")
    return _DcnFn.apply(input.contiguous(), offset.contiguous(), mask.contiguous(), weight, bias, int(deform_groups))


# ------------------------------------------------------------------------------------------ predictor pieces
class _GConvFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, cpg, act, slope):
        out = ops.gconv3x3(x, weight, bias, cpg, act, slope)
        ctx.meta = (cpg, act, slope)
        ctx.params = [weight, bias]      # the caller's tensor objects (grad_sink keys on them)
        ctx.save_for_backward(x, weight, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        cpg, act, slope = ctx.meta
        x, weight, out = ctx.saved_tensors
        g = ops.act_bwd(dout.contiguous(), out, act, slope) if act is not None else dout.contiguous()
        if ctx.params[1] is not None and all(ctx.needs_input_grad[1:3]) and grad_sink.eligible(ctx.params):
            # the uses of the front end's weights across the frames add into one buffer (was: one ATen add per use and tensor)
            bufs, acc = grad_sink._active.raw(ctx.params)
            dx = ops.gconv3x3_bwd(g, x, weight, cpg, grads=tuple(bufs), accumulate=acc)[0]
            return dx, None, None, None, None, None
        dx, dw, db = ops.gconv3x3_bwd(g, x, weight, cpg)
        return dx, dw, db, None, None, None


def adapt_frontend(x, h_hr, w1, b1, w2, b2):
    if not _needs_grad(x, h_hr, w1, b1, w2, b2):
        return ops.adapt_frontend(x, h_hr, w1, b1, w2, b2)
    t = torch.cat([x, h_hr], 1)                                   # networks.py:300 / :336, un-fused for training
    t = _GConvFn.apply(t, w1, b1, 1, "lrelu", 0.2)
    return _GConvFn.apply(t, w2, b2, 2, "lrelu", 0.2)


class _AffineFn(Function):
    @staticmethod
    def forward(ctx, heads, D, with_mask):
        off, mask = ops.affine_offsets(heads, D, with_mask)
        ctx.D = D
        ctx.save_for_backward(*( [mask] if with_mask else []))
        if with_mask:
            return off, mask
        return off

    @staticmethod
    def backward(ctx, doff, dmask=None):
        mask = ctx.saved_tensors[0] if ctx.saved_tensors else None
        return ops.affine_offsets_bwd(doff.contiguous(), None if dmask is None else dmask.contiguous(), mask, ctx.D), None, None


def affine_offsets(heads, D, with_mask):
    if not _needs_grad(heads):
        return ops.affine_offsets(heads, D, with_mask)
    r = _AffineFn.apply(heads, D, with_mask)
    return r if with_mask else (r, None)


# ------------------------------------------------------------------------------------------ resampling glue
class _ResizeFn(Function):
    @staticmethod
    def forward(ctx, x, pre_add, post_add, size, scale):
        ctx.meta = (tuple(x.shape), scale)
        return ops.resize_bilinear_ac(x, size, scale, pre_add=pre_add, post_add=post_add)

    @staticmethod
    def backward(ctx, dout):
        shape, scale = ctx.meta
        dout = dout.contiguous()
        din = ops.resize_bilinear_ac_bwd(dout, shape, scale) if (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) else None
        return (din if ctx.needs_input_grad[0] else None, din if ctx.needs_input_grad[1] else None,
                dout if ctx.needs_input_grad[2] else None, None, None)


def resize_bilinear_ac(x, size, scale=1.0, pre_add=None, post_add=None):
    if not _needs_grad(x, pre_add, post_add):
        return ops.resize_bilinear_ac(x, size, scale, pre_add=pre_add, post_add=post_add)
    return _ResizeFn.apply(x, pre_add, post_add, (int(size[0]), int(size[1])), float(scale))


class _PyramidFn(Function):
    @staticmethod
    def forward(ctx, x):
        return ops.pyramid(x)

    @staticmethod
    def backward(ctx, dd2, dd4):
        return ops.pyramid_bwd(dd2.contiguous(), dd4.contiguous())


def pyramid(x):
    return _PyramidFn.apply(x) if _needs_grad(x) else ops.pyramid(x)


class _AddFn(Function):
    @staticmethod
    def forward(ctx, a, b, c):
        return ops.add(a, b, c)

    @staticmethod
    def backward(ctx, d):
        return d, d, (d if ctx.needs_input_grad[2] else None)


def add(a, b, c=None, out=None):
    """`out` (inference only): write into an existing tensor; ignored when a gradient is needed"""
    return _AddFn.apply(a, b, c) if _needs_grad(a, b, c) else ops.add(a, b, c, out=out)


# ------------------------------------------------------------------------------------------ RCAB tail
class _RcabTailFn(Function):
    """out = r * sigmoid(W2 relu(W1 mean_hw(r) + b1) + b2) + x   (CALayer + residual, networks.py:444-447,463-464)"""

    @staticmethod
    def forward(ctx, r, x, w1, b1, w2, b2):
        n, c, h, w = r.shape
        mean = ops.plane_sum(r, None, 1.0 / (h * w))
        scale = ops.ca_scale(mean.view(n, 1, c), 1, w1, b1, w2, b2)
        ctx.save_for_backward(r, mean, scale, w1, b1, w2, b2)
        ctx.params = [w1, b1, w2, b2]      # the caller's tensor objects (grad_sink keys on them)
        return ops.scale_residual(r, scale, x)

    @staticmethod
    def backward(ctx, d):
        r, mean, scale, w1, b1, w2, b2 = ctx.saved_tensors
        d = d.contiguous()
        n, c, h, w = r.shape
        if ops.rcab_tail_bwd_supported(c, int(w1.shape[0])):
            # plane sums + ONE launch; inside grad_sink the four parameter gradients are added in place by that launch
            if all(ctx.needs_input_grad[2:]) and grad_sink.eligible(ctx.params):
                bufs, acc = grad_sink._active.raw(ctx.params)
                dr = ops.rcab_tail_bwd(d, r, mean, scale, w1, b1, w2, b2, grads=tuple(bufs), accumulate=acc)[0]
                return dr, d, None, None, None, None
            dr, dw1, db1, dw2, db2 = ops.rcab_tail_bwd(d, r, mean, scale, w1, b1, w2, b2)
            return dr, d, dw1, db1, dw2, db2
        dscale = ops.plane_sum(d, r)
        dmean, dw1, db1, dw2, db2 = ops.ca_mlp_bwd(mean, w1, b1, w2, b2, dscale)
        dr = ops.scale_residual_bwd(d, scale, dmean * (1.0 / (h * w)))
        return dr, d, dw1, db1, dw2, db2


def rcab_tail(r, x, w1, b1, w2, b2):
    return _RcabTailFn.apply(r, x, w1, b1, w2, b2)


# ------------------------------------------------------------------------------------------ whole RCAB
class _RcabFn(Function):
    """out = x + r * sigmoid(W_b relu(W_a mean_hw(r) + b_a) + b_b),  r = conv3x3(relu(conv3x3(x)))   (RCABlock.forward,
    networks.py:461-464, mode 'CRC') as ONE autograd node for the training step (round 5): the forward takes the channel
    sums from the second convolution's epilogue (no plane-sum launch), the backward hands the residual path's gradient to the
    last input-gradient convolution as its `residual` (no ATen addition by autograd), and the weight / bias gradients of both
    convolutions and the four channel-attention parameters go through grad_sink when it is active."""

    @staticmethod
    def forward(ctx, x, r_prev, w1, b1, w2, b2, wa, ba, wb, bb):
        n, c, h, w = x.shape
        t = ops.conv2d([x], [w1], [b1], act="relu")
        r, partial = ops.conv2d([t], [w2], [b2], chan_partial=True)
        # the tail as ONE launch (every workgroup redoes the 64 -> 4 -> 64 MLP of its sample from the per-tile sums: 9 KB at a crop;
        # in the one-stream training graph the launch it saves is a pure gain -- unlike the two-stream inference step)
        out, scale, mean = ops.ca_tail(r, partial, wa, ba, wb, bb, x, with_stats=True)
        ctx.save_for_backward(x, t, r, mean, scale, w1, w2, wa, ba, wb, bb, r_prev)
        ctx.params = ([w1], [b1], [w2], [b2], [wa, ba, wb, bb])      # the caller's tensor objects (grad_sink keys on them)
        ctx.mark_non_differentiable(r)
        ctx.set_materialize_grads(False)      # (autograd otherwise fills a zero gradient of r's size for the second output: 845 fills per step)
        return out, r

    @staticmethod
    def backward(ctx, d, _dr_unused=None):
        x, t, r, mean, scale, w1, w2, wa, ba, wb, bb, r_prev = ctx.saved_tensors
        pw1, pb1, pw2, pb2, pca = ctx.params
        if d is None:      # (the block's output reached no loss term)
            return (None,) * 10
        d = d.contiguous()
        sink = grad_sink._active if grad_sink.eligible(pw1 + pb1 + pw2 + pb2 + pca) else None
        # tail: the plane sums of d * r, then one launch (MLP backward, mean broadcast, dr; the mean came out of the forward's
        # ca_scale).  Round 6: when d is what the NEXT block's backward returned (the same tensor object: nothing was added to it on
        # the way), that block's last convolution already left the sums in per-tile rows (_chain below): no plane-sum launch
        dsum = _chain.take_sums(d)
        if sink is not None:
            bufs, acc = sink.raw(pca)
            dr = ops.rcab_tail_bwd(d, r, mean, scale, wa, ba, wb, bb, grads=tuple(bufs), accumulate=acc, dsum=dsum)[0]
            dca = (None, None, None, None)
        else:
            dr, *dca = ops.rcab_tail_bwd(d, r, mean, scale, wa, ba, wb, bb, dsum=dsum)
        # second convolution
        if sink is not None:
            sink.add_use(pw2, pb2, 3, dr, [t])
            dW2 = db2 = None
        else:
            dW2, db2 = ops.conv_wgrad(dr, [t], 3), ops.channel_sum(dr)
        # the ReLU's backward mask (t > 0) leaves in the epilogue of conv-2's input-gradient convolution
        g1 = ops.conv2d(dr, w2, None, act="relu_mask", residual=t, dgrad=True)
        # first convolution; the residual path's gradient d rides in its input-gradient convolution
        if sink is not None:
            sink.add_use(pw1, pb1, 3, g1, [x])
            dW1 = db1 = None
        else:
            dW1, db1 = ops.conv_wgrad(g1, [x], 3), ops.channel_sum(g1)
        if r_prev is not None and RCAB_CHAIN:      # x is the previous block's output: its backward starts with sum_hw dx * r_prev
            dx, rows = ops.conv2d(g1, w1, None, residual=d, dgrad=True, sum_mul=r_prev)
            _chain.put_sums(dx, rows)
        else:
            dx = ops.conv2d(g1, w1, None, residual=d, dgrad=True)
        return (dx, None, dW1, db1, dW2, db2) + tuple(dca)


def rcab_supported(x, params) -> bool:
    """the one-node RCAB: 64 channels, every parameter and the input trainable (anything else: the per-op nodes)"""
    return (x.dim() == 4 and int(x.shape[1]) == 64 and x.requires_grad and all(p is not None and p.requires_grad for p in params)
            and tuple(params[0].shape) == (64, 64, 3, 3) and tuple(params[2].shape) == (64, 64, 3, 3)
            and ops.rcab_tail_bwd_supported(64, int(params[4].shape[0])))


# EAVSR_RCAB_CHAIN=0: every block's backward takes its own plane sums (the A/B switch)
RCAB_CHAIN = os.environ.get("EAVSR_RCAB_CHAIN", "1") != "0"


class _Chain:
    """What consecutive RCABs hand each other around autograd (round 6).  Forward: block k's `r` for block k + 1, found by the
    IDENTITY of the tensor object that is block k's output and block k + 1's input.  Backward: the per-tile sums of dx * r_prev
    that block k + 1's last input-gradient convolution left in its epilogue, found by the identity of the tensor that is its
    returned dx and block k's incoming d.  Both tables hold strong references, so an id() cannot be reused while its entry lives;
    an entry whose object is not the one asked about (autograd added another gradient to dx, a hook replaced it) is never used --
    the consumer then takes its plane sums itself.  Entries are dropped when they are consumed and by clear()."""

    def __init__(self):
        self._r_of_out = {}
        self._sums_of_dx = {}

    def put_r(self, out, r):
        if len(self._r_of_out) > 64:      # (forwards without a backward, over and over: only a group's LAST block stays unconsumed)
            self._r_of_out.clear()
        self._r_of_out[id(out)] = (out, r)

    def r_behind(self, x):
        ent = self._r_of_out.pop(id(x), None)
        return ent[1] if ent is not None and ent[0] is x else None

    def put_sums(self, dx, rows):
        if len(self._sums_of_dx) > 64:
            self._sums_of_dx.clear()
        self._sums_of_dx[id(dx)] = (dx, rows)

    def take_sums(self, d):
        ent = self._sums_of_dx.pop(id(d), None)
        return ent[1] if ent is not None and ent[0] is d else None

    def clear(self):
        self._r_of_out.clear()
        self._sums_of_dx.clear()


_chain = _Chain()


def rcab(x, w1, b1, w2, b2, wa, ba, wb, bb):
    r_prev = _chain.r_behind(x) if RCAB_CHAIN else None
    if r_prev is not None and r_prev.shape != x.shape:
        r_prev = None
    out, r = _RcabFn.apply(x, r_prev, w1, b1, w2, b2, wa, ba, wb, bb)
    if RCAB_CHAIN:
        _chain.put_r(out, r)
    return out


# the remaining ops have no trainable use on the path: forwarded as is
selftest_mfma = ops.selftest_mfma
ca_scale = ops.ca_scale
scale_residual = ops.scale_residual
ca_fusable = ops.ca_fusable
profile = ops.profile
lib = ops.lib
needs_grad = _needs_grad
