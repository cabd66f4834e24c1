"""Multi-GPU mode: independent stereo pairs are sharded across ranks (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).  A Frame's features depend only on its own
two images (reference src/Frame.cc:136-221), so there is NO data-path collective; the only communication is the
gather of every rank's trimmed feature record (records.py / csrc/records.hip) to rank 0: point-to-point
sends, so that every peer uses its own xGMI link into rank 0.  The record sizes travel as a one-word all_gather issued when the record is packed
(SizeExchange) and are read when it is sent, one step later in a pipelined run: no host wait for the device on the way.
"""
import numpy as np


def shard_range(n_frames, rank, world):
    """Contiguous block [lo, hi) of frames for `rank`; blocks differ by at most one frame and tile [0, n_frames)."""
    base, rem = divmod(int(n_frames), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class SizeExchange:
    """The record sizes of all ranks, asked for when a step's record is packed and read when it is sent -- one step later in a pipelined run, so the host never
    waits for the device on the way: the packer's byte counter (a device word) goes into an all_gather on the communication stream as it is, the result into
    pinned host memory behind it.  `sizes(cap)` returns the ranks' byte counts (clamped to the send buffers' capacity).  gloo (host tensors only) reads the
    counter at once."""

    def __init__(self, nbytes_dev, dist, stream=None):
        import torch
        self.world = dist.get_world_size()
        self.event = None
        if dist.get_backend() == "gloo" or not nbytes_dev.is_cuda:
            out = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world)]
            dist.all_gather(out, nbytes_dev.detach().to("cpu", torch.int64).reshape(1))
            self.host = torch.cat(out)
            return
        stream = stream if stream is not None else torch.cuda.current_stream(nbytes_dev.device)
        stream.wait_stream(torch.cuda.current_stream(nbytes_dev.device))
        with torch.cuda.stream(stream):
            gathered = torch.empty(self.world, dtype=torch.int64, device=nbytes_dev.device)
            dist.all_gather_into_tensor(gathered, nbytes_dev.reshape(1).to(torch.int64))
            self.host = torch.empty(self.world, dtype=torch.int64, pin_memory=True)
            self.host.copy_(gathered, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record(stream)
            self._keep = gathered

    def sizes(self, cap=None):
        if self.event is not None:
            self.event.synchronize()
        v = [int(x) for x in self.host.tolist()]
        return [min(x, int(cap)) for x in v] if cap is not None else v


def gather_records(packed, nbytes, dist, dst=0, recv=None, sizes=None):
    """Gather the first `nbytes` bytes of every rank's uint8 tensor `packed` to rank `dst`.

    Returns (records, sizes): on rank dst `records` is the list of the ranks' records in rank order (its own one is a view of `packed`),
    elsewhere None; `sizes` are all ranks' byte counts.  recv: optional list of preallocated uint8 receive tensors (one per rank) so that
    a steady-state step allocates nothing.  sizes: all ranks' byte counts when the caller already has them (SizeExchange) -- `nbytes` is then
    ignored and no size exchange takes place here."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = packed.device
    if sizes is not None:
        nbytes = sizes[rank]
    if dist.get_backend() == "gloo" and packed.is_cuda:
        # gloo moves host memory only: the same size exchange and point-to-point sends on a host copy of the record (the one-GPU test of the
        # N > 1 path, bench.py --backend gloo; on a multi-GPU node the backend is RCCL and the device buffers go out as they are)
        host = packed[:int(nbytes)].cpu()
        recs, sizes = gather_records(host, nbytes, dist, dst, None, sizes)
        if recs is not None:
            for r in range(world):
                if r == dst:
                    recs[r] = packed[:sizes[r]]
                elif recv is not None:
                    recv[r][:sizes[r]].copy_(recs[r]); recs[r] = recv[r][:sizes[r]]
                else:
                    recs[r] = recs[r].to(dev)
        return recs, sizes
    if sizes is None:
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
