"""HIP-graph replay of a whole forward or training step (one process, one GPU, fixed input shape).

`EAVSRP.forward` is ~3,400 kernel launches per 4 x 7 x 180 x 320 step; the host needs 170-190 ms of Python / ctypes
time to enqueue them against 370 ms of device time (tools/gpu_enqueue_time.py).  That is hidden today, but it is the
next bound once the kernels get faster, and it is host time that eight ranks on one node compete for.  A captured HIP
graph replays the same launches in about a millisecond of host time.

Every kernel of libeavsr_hip.so is launched on `torch.cuda.current_stream()` and allocates nothing itself, so the standard
`torch.cuda.graph` capture works unchanged: outputs and temporaries live in the graph's private memory pool, the input
is a static buffer that `__call__` copies into.  One-time work (weight packing, hipFuncSetAttribute) happens in the
warm-up runs before the capture.
"""
from __future__ import annotations

import os

import torch


class GraphedForward:
    """Capture `module(example)` under `torch.no_grad()` and replay it for inputs of the same shape / dtype.

    The returned tensor is the graph's static output buffer: it is overwritten by the next call (clone it to keep it).
    Weights are read in place, so `load_state_dict` / in-place updates are seen by later replays -- but the packed
    forms of the weights are cached per parameter version outside the graph, so call `recapture()` after changing them.
    """

    def __init__(self, module, example: torch.Tensor, warmup: int = 2):
        if not example.is_cuda:
            raise RuntimeError("GraphedForward needs a CUDA (HIP) tensor: eavsr_amd has no CPU path")
        self.module = module
        self.static_in = example.clone()
        self.warmup = warmup
        self.graph = None
        self.static_out = None
        self.recapture()

    def recapture(self):
        side = torch.cuda.Stream(device=self.static_in.device)
        side.wait_stream(torch.cuda.current_stream(self.static_in.device))
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(self.warmup):        # packs weights, sets kernel attributes, warms the allocator
                self.module(self.static_in)
        torch.cuda.current_stream(self.static_in.device).wait_stream(side)
        torch.cuda.synchronize(self.static_in.device)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: a collective library's watchdog thread (RCCL, when a process group exists) must not invalidate the capture
        with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.static_out = self.module(self.static_in)
        # the captured launches read the derived weight forms (packed / Winograd-transformed) that the warm-up left in the
        # per-parameter caches: hold them, so that clearing or refreshing a cache cannot free memory the graph still uses
        from . import ops
        self._weights_alive = [dict(getattr(d, "_d", d)) for d in ops.WEIGHT_CACHES]

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape != self.static_in.shape or x.dtype != self.static_in.dtype or x.device != self.static_in.device:
            raise ValueError(f"graph captured for {tuple(self.static_in.shape)} {self.static_in.dtype} on "
                             f"{self.static_in.device}, got {tuple(x.shape)} {x.dtype} on {x.device}")
        self.static_in.copy_(x)
        self.graph.replay()
        return self.static_out



class StreamedForward:
    """The clips of a step as `groups` independent sub-batches, each captured as its own HIP graph and replayed on its
    own stream.

    Clips are independent and the step is a chain of kernels with complementary bottlenecks: the Winograd
    convolutions keep the matrix pipe and all of a CU's LDS but little HBM bandwidth, `scale_residual`, the warps and the
    other streaming kernels the opposite.  With two sub-batches in flight one group's streaming kernels run in the wave
    slots the other group's convolution leaves free.  Graphs are required: two eager forwards double the launches and
    the host would become the bound.  Results are bit-identical to the single-stream forward (same kernels per clip).
    """

    # Start-up skew between the streams, microseconds per stream index (stream g waits g x skew behind a device-side delay before
    # its replay).  Both sub-batches run the SAME program; started together they stay in phase -- both in an alignment step
    # (small streaming kernels that cannot fill the chip) at the same time, then both in the 62 convolutions of a backbone call,
    # which serialise on the CUs (one 132 KB-LDS workgroup each) -- so the alignment steps are exposed twenty-eight times per
    # forward.  Half a backbone call out of phase, one stream's alignment runs under the other's convolutions.  DESIGN.md 3.5.
    SKEW_US = float(os.environ.get("EAVSR_STREAM_SKEW_US", "0"))

    def __init__(self, module, example: torch.Tensor, groups: int = 2, warmup: int = 2, skew_us: float | None = None):
        self.skew_us = self.SKEW_US if skew_us is None else float(skew_us)
        self._ticks_per_us = 0.0
        if self.skew_us > 0:      # what one tick of torch's spin kernel lasts on this device, measured once
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(1_000_000)
            torch.cuda.synchronize(example.device)
            e0.record()
            torch.cuda._sleep(20_000_000)
            e1.record()
            torch.cuda.synchronize(example.device)
            self._ticks_per_us = 20_000_000 / max(1e-3, e0.elapsed_time(e1) * 1e3)
        n = example.shape[0]
        if groups < 1 or n % groups:
            raise ValueError(f"{n} clips do not split into {groups} equal groups")
        self.per = n // groups
        self.streams = [torch.cuda.Stream(device=example.device) for _ in range(groups)]
        self.parts = []
        for g in range(groups):
            self.parts.append(GraphedForward(module, example[g * self.per:(g + 1) * self.per].contiguous(), warmup=warmup))
        self.shape, self.dtype, self.device = example.shape, example.dtype, example.device
        out0 = self.parts[0].static_out
        self.static_out = torch.empty((n,) + tuple(out0.shape[1:]), device=out0.device, dtype=out0.dtype)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape != self.shape or x.dtype != self.dtype or x.device != self.device:
            raise ValueError(f"graphs captured for {tuple(self.shape)} {self.dtype} on {self.device}")
        cur = torch.cuda.current_stream(self.device)
        for g, (part, st) in enumerate(zip(self.parts, self.streams)):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                part.static_in.copy_(x[g * self.per:(g + 1) * self.per])
                if g and self.skew_us > 0:
                    torch.cuda._sleep(int(self.skew_us * g * self._ticks_per_us))
                part.graph.replay()
                self.static_out[g * self.per:(g + 1) * self.per].copy_(part.static_out)
        for st in self.streams:
            cur.wait_stream(st)
        return self.static_out

def clear_weight_caches():
    """Drop every per-(parameter, version) cache of derived weights (packed / transformed / transposed forms).

    A replayed graph updates the parameters on the device without touching their Python-side version counters, so
    a cache filled inside (or before) a capture must not serve eager calls afterwards - and a capture must not hit an
    entry built outside it, or the replays would keep reading that stale copy."""
    from . import autograd, ops      # (importing autograd registers its caches)
    for d in ops.WEIGHT_CACHES:      # every derived-weight cache registers itself there (ops.register_weight_cache)
        d.clear()


class GraphedTrainStep:
    """`EAVSRPModel.optimize_parameters()` (forward, L1 loss, backward, Adam) captured once and replayed.

    The training step of configs[3] is ~14,000 kernel launches of 5-40 us; eager Python needs longer to enqueue them
    than the GPU needs to run them.  One process: the whole step, optimizer included, is one graph.  Data parallel
    (world size > 1): the graph ends after backward; the bucketed all-reduce (`shard.GradientAllReducer.finish`) and the
    Adam step run eagerly after each replay (hooks cannot fire from a replay, so there is no overlap - 49 MB of
    gradients against a 0.4 s step).

    `model.set_input(first_batch)` must have been called; `step(batch)` copies a batch of the same shapes into the
    static buffers and replays.  `model.loss_*` are the graph's static loss tensors.
    """

    def __init__(self, model, warmup: int = 2):
        import torch.distributed as dist
        self.model = model
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        dev = model.device
        if model.data_hr_seq is None:
            raise RuntimeError("GraphedTrainStep: call model.set_input() with lr_seq and hr_seq first")
        self.static_lr = model.data_lr_seq.clone()
        self.static_hr = model.data_hr_seq.clone()
        opt = model.optimizer_EAVSRP
        self._capturable_before = [(g.get("capturable", False), g.get("fused", None), g.get("foreach", None)) for g in opt.param_groups]
        # Adam keeps `step` on the device and never reads it back (capturable), and runs as torch's FUSED multi-tensor kernel: the
        # default (foreach) capturable form with a device-tensor learning rate divides two 0-dim tensors PER PARAMETER -- 2,432 launches
        # of 4 us in a 196 ms step (profiles/r05_rocprof_summary_train.txt before this change).  EAVSR_ADAM_FUSED=0: A/B switch.
        self._fused = os.environ.get("EAVSR_ADAM_FUSED", "1") == "1"
        for g in opt.param_groups:
            g["capturable"] = True
            if self._fused:
                g["fused"], g["foreach"] = True, False
        for st in opt.state.values():
            if "step" in st and not st["step"].is_cuda:
                st["step"] = st["step"].to(dev, torch.float32)
        model.data_lr_seq, model.data_hr_seq = self.static_lr, self.static_hr
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):          # allocator warm-up, kernel attributes, optimizer state
                model.optimize_parameters()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        clear_weight_caches()
        # nothing may keep the warm-up's autograd graph (and its AccumulateGrad nodes, bound to the side stream) alive
        model.loss_EAVSRP_L1 = model.loss_EAVSRP_Total = model.data_sr_seq = model.data_sr = None
        self.graph = torch.cuda.CUDAGraph()
        model.grad_sync.paused = True
        # A captured Adam bakes a Python-float learning rate into its kernels' arguments: later changes of
        # `param_groups[i]['lr']` (model.update_learning_rate(), base_model.py:131-138) would be ignored by every replay.
        # The capture therefore sees each group's rate as a DEVICE tensor (torch's capturable Adam reads it on the device);
        # afterwards the groups hold their Python floats again, for the schedulers, and `step()` copies them into the tensors.
        self._lr_dev = [torch.tensor(float(g["lr"]), device=dev, dtype=torch.float32) for g in opt.param_groups]
        self._lr_seen = [float(g["lr"]) for g in opt.param_groups]
        try:
            opt.zero_grad(set_to_none=True)
            if self.world == 1:
                for g, lr_t in zip(opt.param_groups, self._lr_dev):
                    g["lr"] = lr_t
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                model.forward()
                self._backward()
                if self.world == 1:
                    opt.step()
        finally:
            for g, lr in zip(opt.param_groups, self._lr_seen):
                g["lr"] = lr
            model.grad_sync.paused = False
            clear_weight_caches()
        # the replayed backward writes into THESE gradient tensors (graph-pool memory).  An eager
        # `model.optimize_parameters()` in between rebinds `p.grad` (zero_grad(set_to_none=True)); `step()` binds them back
        # so that the all-reduce and Adam that follow a replay never see stale eager gradients.
        self._grads = [(p, p.grad) for g in opt.param_groups for p in g["params"] if p.grad is not None]

    def close(self):
        """Give the optimizer back its pre-capture flags (`capturable`) and drop the graph."""
        for g, (c, fu, fe) in zip(self.model.optimizer_EAVSRP.param_groups, self._capturable_before):
            g["capturable"] = c
            if self._fused:
                g["fused"], g["foreach"] = fu, fe
        self.graph = None
        self._grads = []

    def _backward(self):
        from . import autograd as AG
        m = self.model
        m.loss_EAVSRP_L1 = (m.data_hr_seq - m.data_sr_seq).abs().mean()
        m.loss_EAVSRP_Total = m.loss_EAVSRP_L1
        with AG.grad_sink():
            m.loss_EAVSRP_Total.backward()

    def step(self, batch=None):
        m = self.model
        if batch is not None:
            lr, hr = batch["lr_seq"], batch["hr_seq"]
            if lr.shape != self.static_lr.shape or hr.shape != self.static_hr.shape:
                raise ValueError(f"graph captured for lr {tuple(self.static_lr.shape)} / hr {tuple(self.static_hr.shape)}")
            self.static_lr.copy_(lr)
            self.static_hr.copy_(hr)
        if self.graph is None:
            raise RuntimeError("GraphedTrainStep: closed")
        m.data_lr_seq, m.data_hr_seq = self.static_lr, self.static_hr
        for p, g in self._grads:
            p.grad = g
        if self.world == 1:      # the replayed Adam reads its learning rates from these device scalars
            for i, g in enumerate(m.optimizer_EAVSRP.param_groups):
                lr = float(g["lr"])
                if lr != self._lr_seen[i]:
                    self._lr_dev[i].fill_(lr)
                    self._lr_seen[i] = lr
        self.graph.replay()
        if self.world == 1:
            # the replayed Adam has changed every parameter without bumping its `_version`: whatever an eager call between two
            # replays (an evaluation pass, an eager backward) cached for this version is stale now.  In a pure replay
            # loop the caches are empty and this is sixteen no-ops.
            clear_weight_caches()
        if self.world > 1:
            m.grad_sync.reset()
            m.grad_sync.finish()
            m.optimizer_EAVSRP.step()
