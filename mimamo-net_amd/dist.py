"""Multi-GPU scaling: videos are independent units, so they shard across ranks with NO collective on the
data path.  One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on
ROCm; "gloo" in the CPU tests).  The only exchange is the final gather of [frames, 2] fp32 results
(8 bytes per frame) so every rank -- or just rank 0 -- can assemble the per-video tables.

The reference is single-process / single-device (api/steerable/utils.py:34-50); this module is the build's
addition, not a port of anything.
"""
import os

import numpy as np
import torch


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, world, local_rank)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard(n_videos, rank, world, lengths=None):
    """Indices of the videos rank `rank` processes.  Equal-length clips: round-robin (c mod world == rank).
    With `lengths`: longest-first greedy onto the least-loaded rank (deterministic, identical on all ranks)."""
    if lengths is None:
        return list(range(rank, n_videos, world))
    order = sorted(range(n_videos), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world
    mine = []
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        load[r] += int(lengths[i])
        if r == rank:
            mine.append(i)
    return sorted(mine)


def gather_rows(local_rows, world, rank, max_rows=None):
    """All-gather variable-length [n_r, C] float tensors; returns the list of per-rank tensors (on every rank)."""
    if world == 1:
        return [local_rows]
    import torch.distributed as dist
    n = torch.tensor([local_rows.shape[0]], dtype=torch.int64, device=local_rows.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    cap = max(counts) if max_rows is None else max_rows
    pad = torch.zeros((cap, local_rows.shape[1]), dtype=local_rows.dtype, device=local_rows.device)
    pad[: local_rows.shape[0]] = local_rows
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return [b[:c] for b, c in zip(bufs, counts)]


def run_sharded(video_lengths, compute_rows, rank, world, device="cpu"):
    """Shard videos, run `compute_rows(video_indices) -> [sum(len), C] tensor` locally, gather, and return
    {video index: [len, C] numpy array} for ALL videos on every rank."""
    mine = shard(len(video_lengths), rank, world, video_lengths)
    rows = compute_rows(mine)
    parts = gather_rows(rows.to(device), world, rank)
    out = {}
    for r, part in enumerate(parts):
        part = part.cpu().numpy()
        off = 0
        for i in shard(len(video_lengths), r, world, video_lengths):
            out[i] = part[off: off + video_lengths[i]]
            off += video_lengths[i]
        assert off == part.shape[0]
    return out
