"""Multi-GPU execution of the hot path: one process per GPU, clips sharded by sequence.

The reference's only parallelism is single-process nn.DataParallel (models/networks.py:67-74):
scatter `lrs` on dim 0, replicate the parameters every forward, gather the outputs.  Clips are
independent units, so the MI355X-native form has NO data-path collective for
inference: rank r owns clips r, r+G, r+2G, ... and keeps its own resident copy of the weights.  The
only communication is control-plane (a barrier and a MAX over per-rank wall times for
measurement), which runs over RCCL when the process group is 'nccl' and over gloo in the CPU tests.
(Training would add one bucketed all-reduce over the 49.1 MB of gradients; it needs the backward
kernels and is not part of this round.)
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
            # RCCL ("nccl") on GPUs; EAVSR_DIST_BACKEND=gloo lets several ranks share one GPU in tests
            backend = os.environ.get("EAVSR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a host scalar (per-rank elapsed time -> job time)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64,
                     device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


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
