"""Multi-GPU mode: independent stereo pairs are sharded across ranks (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).  A Frame's features depend only on its own
two images (reference src/Frame.cc:136-221), so there is NO data-path collective; the only communication is the
gather of every rank's trimmed feature record (records.py / csrc/records.hip) to rank 0: point-to-point
sends, so that every peer uses its own xGMI link into rank 0, preceded by a one-word all_gather of the record sizes.
"""
import numpy as np


def shard_range(n_frames, rank, world):
    """Contiguous block [lo, hi) of frames for `rank`; blocks differ by at most one frame and tile [0, n_frames)."""
    base, rem = divmod(int(n_frames), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_records(packed, nbytes, dist, dst=0, recv=None):
    """Gather the first `nbytes` bytes of every rank's uint8 tensor `packed` to rank `dst`.

    Returns (records, sizes): on rank dst `records` is the list of the ranks' records in rank order (its own one is a view of `packed`),
    elsewhere None; `sizes` are all ranks' byte counts.  recv: optional list of preallocated uint8 receive tensors (one per rank) so that
    a steady-state step allocates nothing."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = packed.device
    if dist.get_backend() == "gloo" and packed.is_cuda:
        # gloo moves host memory only: the same size exchange and point-to-point sends on a host copy of the record (the one-GPU test of the
        # N > 1 path, bench.py --backend gloo; on a multi-GPU node the backend is RCCL and the device buffers go out as they are)
        host = packed[:int(nbytes)].cpu()
        recs, sizes = gather_records(host, nbytes, dist, dst, None)
        if recs is not None:
            for r in range(world):
                if r == dst:
                    recs[r] = packed[:sizes[r]]
                elif recv is not None:
                    recv[r][:sizes[r]].copy_(recs[r]); recs[r] = recv[r][:sizes[r]]
                else:
                    recs[r] = recs[r].to(dev)
        return recs, sizes
    sizes_t = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes_t, torch.tensor([int(nbytes)], dtype=torch.int64, device=dev))
    sizes = [int(s.item()) for s in sizes_t]
    if rank == dst:
        out, ops = [None] * world, []
        out[dst] = packed[:sizes[dst]]
        for r in range(world):
            if r == dst:
                continue
            buf = recv[r][:sizes[r]] if recv is not None else torch.empty(sizes[r], dtype=torch.uint8, device=dev)
            out[r] = buf
            if sizes[r]:
                ops.append(dist.P2POp(dist.irecv, buf, r))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return out, sizes
    if sizes[rank]:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, packed[:sizes[rank]], dst)]):
            w.wait()
    return None, sizes
