"""Clip-level data parallelism: the only way the path shards.

Windows of one clip are serially dependent (window c is seeded by window c-1) and steps within a window are a
Markov chain, so nothing inside a clip can be split; clips are fully independent.  One process per GPU
(`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests), clip c -> rank c % world,
full weight replica per rank (14-40 MB), and exactly ONE exchange at the very end: the finished poses are gathered
to rank 0 (1.42 MB per ZEGGS clip -- far below what ring/tree tuning could matter for).  No collective ever runs
inside the step loop.  (The reference has no distributed path at all: SURVEY s2.2.)
"""
from __future__ import annotations

import numpy as np


def shard_clips(n_clips: int, rank: int, world: int) -> list[int]:
    """Round-robin assignment: the clip indices rank `rank` samples."""
    return list(range(rank, n_clips, world))


def gather_poses(local, n_clips: int, dist=None, dst: int = 0, device=None):
    """local: float32 [n_local, F, J] (numpy or torch) holding this rank's clips in shard_clips order.
    Returns [n_clips, F, J] ordered by clip index on rank `dst`, None elsewhere.  Ranks may hold different counts
    (n_clips not divisible by world): shorter shards are padded to the longest for the collective."""
    import torch
    if dist is None or not dist.is_initialized():
        return np.asarray(local.cpu() if hasattr(local, "cpu") else local)
    rank, world = dist.get_rank(), dist.get_world_size()
    t = local if hasattr(local, "cpu") else torch.from_numpy(np.ascontiguousarray(local))
    if device is not None:
        t = t.to(device)
    n_max = (n_clips + world - 1) // world
    F, J = int(t.shape[1]), int(t.shape[2])
    pad = torch.zeros((n_max, F, J), dtype=torch.float32, device=t.device)
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    out = np.zeros((n_clips, F, J), dtype=np.float32)
    for r in range(world):
        idx = shard_clips(n_clips, r, world)
        out[idx] = bufs[r][: len(idx)].cpu().numpy()
    return out
