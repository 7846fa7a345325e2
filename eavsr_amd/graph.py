"""HIP-graph replay of a whole forward (one process, one GPU, fixed input shape).

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
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out = self.module(self.static_in)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape != self.static_in.shape or x.dtype != self.static_in.dtype or x.device != self.static_in.device:
            raise ValueError(f"graph captured for {tuple(self.static_in.shape)} {self.static_in.dtype} on "
                             f"{self.static_in.device}, got {tuple(x.shape)} {x.dtype} on {x.device}")
        self.static_in.copy_(x)
        self.graph.replay()
        return self.static_out
