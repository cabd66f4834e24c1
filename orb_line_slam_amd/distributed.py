"""Multi-GPU mode: independent stereo pairs are sharded across ranks (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).  A Frame's features depend only on its own
two images (reference src/Frame.cc:136-221), so there is NO data-path collective; the only communication is the
gather of the fixed-capacity feature records to rank 0.
"""
import numpy as np


def shard_range(n_frames, rank, world):
    """Contiguous block [lo, hi) of frames for `rank`; blocks differ by at most one frame and tile [0, n_frames)."""
    base, rem = divmod(int(n_frames), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_to_rank0(local, dist, n_frames=None, dst=0):
    """Gather per-frame record tensors to rank `dst`.

    local: dict name -> tensor whose first dimension is this rank's frame count (all ranks use the same names,
    dtypes and trailing shapes).  Returns, on rank dst, dict name -> tensor with first dimension n_frames (the
    concatenation in rank order, i.e. global frame order for shard_range blocks); None elsewhere."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    names = sorted(local)
    n_local = int(next(iter(local.values())).shape[0]) if names else 0
    dev = next(iter(local.values())).device if names else torch.device("cpu")
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([n_local], dtype=torch.int64, device=dev))
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes) if sizes else 0
    out = {}
    for name in names:
        t = local[name].contiguous()
        if t.shape[0] < mx:   # pad to the largest shard so that a plain gather can be used
            pad = torch.zeros((mx - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            t = torch.cat([t, pad], 0)
        bufs = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
        dist.gather(t, bufs, dst=dst)
        if rank == dst:
            out[name] = torch.cat([b[:sizes[r]] for r, b in enumerate(bufs)], 0)
    if rank != dst:
        return None
    if n_frames is not None:
        assert all(v.shape[0] == n_frames for v in out.values())
    return out


def gather_counts(counts, lcounts, dist):
    """bench.py helper: exercise the gather on the per-image count vectors."""
    return gather_to_rank0({"counts": counts, "lcounts": lcounts}, dist)
