"""Multi-GPU execution of the hot path: one process per GPU, clips sharded by sequence.

The reference's only parallelism is single-process nn.DataParallel (models/networks.py:67-74):
scatter `lrs` on dim 0, replicate the parameters every forward, gather the outputs.  Clips are
independent units, so the MI355X-native form has NO data-path collective for
inference: rank r owns clips r, r+G, r+2G, ... and keeps its own resident copy of the weights.  The
only communication is control-plane (a barrier and a MAX over per-rank wall times for
measurement): a few host scalars, which go over gloo -- the default process group on GPUs is 'cpu:gloo,cuda:nccl', so that a
barrier never enqueues a kernel on the device it brackets and RCCL carries only device tensors (the training step's gradients).
Training is plain data parallelism: every rank holds the full model, runs
forward + backward on its own clips, and the ONLY data-path communication is one all-reduce(sum) of the
loss gradients of the 12.28 M trainable parameters (49.1 MB fp32) per step -- `GradientAllReducer`
below: gradients are packed into ~8 MB buckets in reverse registration order (the order backward
produces them); a bucket's all-reduce is launched asynchronously from a post-accumulate-grad hook once
every one of its parameters has received its gradient through autograd, and `finish()` launches the
remaining buckets, waits, divides by the world size and scatters the result back into `.grad`.
What overlaps in practice: the training step accumulates the convolution weight / bias gradients in
`autograd.grad_sink` buffers that reach `.grad` only when backward ends (no hook fires for them), and
nearly every bucket holds a convolution parameter -- so almost all of the 49 MB is reduced in `finish()`,
AFTER backward, not under it.  That is a deliberate trade: the sink removes ~12,000 launches per step,
while the whole all-reduce is ~0.6 ms at the per-link xGMI bound against a 0.4 s step.  No parameter
broadcast per step (the reference's nn.DataParallel re-broadcasts all 54.9 MB every forward): `broadcast_module` runs ONCE, when
the model is constructed and after a checkpoint is loaded, so that ranks agree whatever each of them seeded or loaded.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process if unset)."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


def clip_indices(n_clips: int, rank: int, world: int) -> List[int]:
    """Round-robin ownership: clip i -> rank i mod world (what DataParallel's scatter does in chunks;
    round-robin keeps ranks within one clip of each other for any n_clips)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} not in [0, {world})")
    return list(range(rank, n_clips, world))


def shard_clips(lrs: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """The (n_local, t, c, h, w) slice of a global batch this rank owns."""
    idx = clip_indices(lrs.shape[0], rank, world)
    return lrs[idx] if idx else lrs[:0]


def init_process_group(backend: str | None = None) -> Tuple[int, int, int]:
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # On GPUs: RCCL ("nccl") for device tensors -- the gradient all-reduce of the training step, the path's only
            # data-path collective -- and gloo for host tensors: the control plane of a measurement (barrier, MAX over the
            # per-rank wall times) is a few host scalars, and keeping it off the GPU means a barrier never enqueues a kernel
            # on the device it brackets (and that inference, which has no collective, never opens a RCCL communicator).
            # EAVSR_DIST_BACKEND=gloo lets several ranks share one GPU in tests; =nccl forces everything onto RCCL.
            backend = os.environ.get("EAVSR_DIST_BACKEND") or ("cpu:gloo,cuda:nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            # one process per GPU: bind this process to its device BEFORE the communicator exists, so that RCCL's
            # collectives (and barrier) never have to guess the device from the rank
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def host_collectives() -> bool:
    """True when host tensors have a backend of their own (gloo): barrier / MAX then run on the CPU"""
    return dist.is_initialized() and "gloo" in str(dist.get_backend())


def barrier():
    if dist.is_initialized():
        if host_collectives():
            dist.all_reduce(torch.zeros(1))                       # a host all-reduce IS a barrier; never touches the GPU
        elif "nccl" in str(dist.get_backend()):
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a host scalar (per-rank elapsed time -> job time)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64,
                     device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ranks(value: float, device=None) -> List[float]:
    """every rank's host scalar, in rank order (per-rank step times beside the MAX the job is quoted on)"""
    if not dist.is_initialized():
        return [float(value)]
    world = dist.get_world_size()
    t = torch.zeros(world, dtype=torch.float64, device=device if device is not None else "cpu")
    t[dist.get_rank()] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]


def all_ranks_str(value: str) -> List[str]:
    """every rank's short string, in rank order (device names; through a fixed-size byte tensor: works on any backend that has
    all_reduce, which is all this module relies on)"""
    if not dist.is_initialized():
        return [value]
    world, rank = dist.get_world_size(), dist.get_rank()
    raw = value.encode()[:96]
    dev = torch.device("cuda", torch.cuda.current_device()) if (not host_collectives() and torch.cuda.is_available()) else "cpu"
    t = torch.zeros(world, 96, dtype=torch.int32, device=dev)
    t[rank, :len(raw)] = torch.tensor(list(raw), dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [bytes(int(b) for b in row if b).decode(errors="replace") for row in t.cpu().tolist()]


def all_ranks_ok(ok: bool) -> bool:
    """True iff EVERY rank passes ok=True: one MIN all-reduce of a flag (host tensor over gloo where the group has it, a device
    tensor otherwise).  The agreement step in front of a start-up collective: a rank that failed locally (a missing checkpoint,
    a shape mismatch) must take the others down with it instead of leaving them blocked in the broadcast that follows, or
    pairing that broadcast with their next collective.  Every rank must call it."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return bool(ok)
    dev = "cpu" if (host_collectives() or not torch.cuda.is_available()) else torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def cap_host_threads(world: int) -> int:
    """N ranks on one node share its cores: cap this rank's intra-op / OpenMP threads at cores // world (at least 1), so that
    eight ranks do not start eight full-width thread pools.  Returns the cap."""
    import os
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    cap = max(1, cores // max(1, world))
    if world > 1:
        torch.set_num_threads(min(torch.get_num_threads(), cap))
    return cap


def sum_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64,
                     device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_outputs(local: torch.Tensor, n_clips: int, rank: int, world: int) -> torch.Tensor | None:
    """Optional: reassemble the global (n_clips, ...) output on rank 0 (DataParallel's gather).
    Not used in the timed path."""
    if not dist.is_initialized():
        return local
    counts = [len(clip_indices(n_clips, r, world)) for r in range(world)]
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, bufs, dst=0)
    if rank != 0:
        return None
    out = torch.empty((n_clips,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = clip_indices(n_clips, r, world)
        out[idx] = bufs[r][:len(idx)]
    return out


def broadcast_module(module: torch.nn.Module, src: int = 0, bucket_bytes: int = 64 << 20) -> int:
    """Every parameter and buffer of `module` takes rank `src`'s value: ONE start-up collective per ~64 MB bucket (the model is
    54.9 MB: one or two broadcasts), never per step -- the reference's nn.DataParallel re-broadcasts the replica every forward
    (models/networks.py:67-74).  Ranks agree afterwards even when only rank `src` loaded a checkpoint or when the ranks were
    seeded differently (until round 5 they agreed only because every rank seeded identically).
    Tensors travel in `state_dict()` order, grouped by dtype, packed flat on the tensors' own device (RCCL for device tensors,
    gloo for host tensors).  Returns the number of bytes broadcast; a no-op (0) on one process.  A COLLECTIVE: every rank of the
    group must call it, at the same point of its program (`EAVSRPModel.load_networks` agrees on success first: `all_ranks_ok`)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    tensors = [t for t in module.state_dict(keep_vars=True).values() if isinstance(t, torch.Tensor) and t.numel() > 0]
    total = 0
    by_type = {}
    for t in tensors:
        by_type.setdefault((t.dtype, t.device), []).append(t)
    with torch.no_grad():
        for (dtype, device), group in by_type.items():
            cur, size = [], 0
            buckets = []
            for t in group:
                cur.append(t)
                size += t.numel() * t.element_size()
                if size >= bucket_bytes:
                    buckets.append(cur)
                    cur, size = [], 0
            if cur:
                buckets.append(cur)
            for bucket in buckets:
                flat = torch.cat([t.detach().reshape(-1) for t in bucket])
                dist.broadcast(flat, src=src)
                o = 0
                for t in bucket:
                    n = t.numel()
                    t.detach().copy_(flat[o:o + n].view_as(t))
                    o += n
                total += flat.numel() * flat.element_size()
    return total


def ranks_seen(device=None) -> int:
    """How many ranks the DATA-PATH backend itself sees: one 4-byte all-reduce of ones on `device` (RCCL when it is a GPU; the
    control plane's gloo group cannot answer for it).  bench.py prints it as `rccl_ranks_seen` before the timed region."""
    if not dist.is_initialized():
        return 1
    t = torch.ones(1, dtype=torch.int32, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


class GradientAllReducer:
    """Bucketed all-reduce of `.grad` (the one collective of the training step).  In the training step as it is run
    (`autograd.grad_sink`: convolution weight / bias gradients reach `.grad` only when backward ends, and nearly every ~8 MB bucket
    holds such a parameter) the 49 MB are reduced in `finish()`, AFTER backward -- there is no overlap worth the name, and none
    is needed: ~0.6 ms at the per-link xGMI bound against a 0.4 s step.  A bucket whose parameters ALL receive their gradient
    through ordinary autograd accumulation is still launched from the post-accumulate-grad hook as soon as it is complete.
    Bucket layouts are fixed at construction and identical on every rank: a parameter without a gradient on this rank
    contributes zeros, never a shorter buffer."""

    def __init__(self, params, bucket_bytes: int = 8 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.buckets: List[List[torch.nn.Parameter]] = []
        cur, size = [], 0
        for p in reversed(self.params):           # backward reaches the last layers first
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        self._pending = [0] * len(self.buckets)
        self._work = []
        self._hooks = []
        self.paused = False          # graph capture: no collective from a hook (eavsr_amd.graph.GraphedTrainStep)
        if self.world > 1:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.reset()

    def reset(self):
        self._pending = [len(b) for b in self.buckets]
        self._work = []

    def _launch(self, i: int):
        # rank-invariant size: every parameter of the bucket has its slot, zero-filled where this rank produced no
        # gradient (ranks that disagreed on which parameters got one would otherwise exchange buffers of different
        # lengths -- a hang or silent corruption)
        bucket = self.buckets[i]
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        handle = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        self._work.append((handle, flat, bucket))

    def _on_grad(self, p):
        if self.paused:
            return
        i = self._bucket_of[id(p)]
        self._pending[i] -= 1
        if self._pending[i] == 0:
            self._launch(i)

    def finish(self):
        """Call after loss.backward(): launches buckets whose parameters received no gradient hook (unused
        parameters), waits for every all-reduce and writes the averaged gradients back."""
        if self.world > 1:
            for i, left in enumerate(self._pending):
                if left > 0:
                    self._pending[i] = 0
                    self._launch(i)
            for handle, flat, bucket in self._work:
                handle.wait()
                flat.div_(self.world)
                o = 0
                for p in bucket:
                    n = p.numel()
                    if p.grad is None:           # no local gradient: it still receives the other ranks' average
                        p.grad = flat[o:o + n].view_as(p).clone()
                    else:
                        p.grad.copy_(flat[o:o + n].view_as(p.grad))
                    o += n
        self.reset()

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
