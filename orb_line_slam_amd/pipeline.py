"""Batched offline mode with host images (BASELINE config C4: recorded sequences, frame-sharded): a double-buffered pipeline that overlaps the
host-to-device copy of batch i+1, the feature path of batch i and the device-to-host copy of batch i-1 on three HIP streams, so that the
PCIe-inclusive rate equals the kernel rate instead of the sum of the three.

    pipe = OfflinePipeline(params, 1242, 375, pairs_per_batch=512)
    for frames in pipe.run(batches):          # batches: iterable of (2*n_pairs, H, W) uint8 arrays, n_pairs <= pairs_per_batch
        ...                                   # frames: StereoFrames (as StereoFrontEnd.frames returns); its arrays are views of pinned
                                              # staging buffers that are reused two batches later -- copy what you keep

PyTorch provides the pinned / device memory and the streams; every computation is the C ABI's olf_stereo_frames_dev.
"""
import ctypes as C
import numpy as np
from . import _lib
from ._lib import DESC_BYTES, KEYLINE_DTYPE, KEYPOINT_DTYPE, FrameBuffers, check, lib
from .frame import StereoFrames


def _gpu_local_cpus(torch, dev):
    """CPUs of the NUMA node the GPU hangs off (sysfs), or None.  Pinned staging memory is placed by first touch: allocated from a CPU of another node, every
    DMA of the pipeline crosses the socket interconnect -- measured on MI355X boxes as 330 ms per 3072-pair batch instead of 220 (profiles/r4am_pcie_numa.txt)."""
    try:
        pr = torch.cuda.get_device_properties(dev)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        txt = open("/sys/bus/pci/devices/%s/local_cpulist" % bdf).read().strip()
        cpus = set()
        for part in txt.split(","):
            if "-" in part:
                a, b = part.split("-"); cpus.update(range(int(a), int(b) + 1))
            elif part:
                cpus.add(int(part))
        return cpus or None
    except Exception:
        return None


class _LocalAffinity:
    """with _LocalAffinity(torch, dev): the calling thread runs on the GPU's NUMA node (first-touch placement of what is allocated inside)"""
    def __init__(self, torch, dev):
        self.cpus = _gpu_local_cpus(torch, dev); self.old = None
    def __enter__(self):
        import os
        try:
            if self.cpus:
                self.old = os.sched_getaffinity(0)
                use = self.cpus & self.old
                if use: os.sched_setaffinity(0, use)
        except Exception:
            self.old = None
        return self
    def __exit__(self, *a):
        import os
        try:
            if self.old: os.sched_setaffinity(0, self.old)
        except Exception:
            pass


class OfflinePipeline:
    def __init__(self, params=None, width=1242, height=375, pairs_per_batch=256, device=None):
        import torch
        self.torch = torch
        self.params = params or _lib.default_params()
        self.width, self.height, self.B = width, height, pairs_per_batch
        import os
        self.input_event = os.environ.get("OLF_PIPE_EVENT", "0") != "0"      # olf_ctx_set_input_event with each batch's upload event (A/B switch)
        self.dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        with torch.cuda.device(self.dev):
            self.ctx = _lib.Context(self.params, width, height, 2 * pairs_per_batch)
            cap, lcap, B = self.ctx.orb_capacity, self.ctx.line_capacity, pairs_per_batch
            self.cap, self.lcap = cap, lcap
            spec = [("kps", (2 * B, cap, 28), torch.uint8), ("desc", (2 * B, cap, DESC_BYTES), torch.uint8), ("counts", (2 * B,), torch.int32),
                    ("ur", (B, cap), torch.float32), ("dp", (B, cap), torch.float32), ("kls", (2 * B, lcap, 68), torch.uint8),
                    ("ldesc", (2 * B, lcap, DESC_BYTES), torch.uint8), ("lcounts", (2 * B,), torch.int32), ("lm", (B, lcap), torch.int32),
                    ("ldisp", (B, lcap, 2), torch.float32), ("lle", (B, lcap, 3), torch.float64)]
            self.names = [s[0] for s in spec]
            self.slots = []
            for _ in range(2):
              with _LocalAffinity(torch, self.dev):      # (the staging buffers on the GPU's NUMA node)
                slot = {"host_in": torch.zeros((2 * B, height, width), dtype=torch.uint8).pin_memory(),
                        "dev_in": torch.empty((2 * B, height, width), dtype=torch.uint8, device=self.dev),
                        "dev": {n: torch.zeros(sh, dtype=dt, device=self.dev) for n, sh, dt in spec},
                        "host": {n: torch.zeros(sh, dtype=dt).pin_memory() for n, sh, dt in spec},
                        "in_done": torch.cuda.Event(), "cmp_done": torch.cuda.Event(), "out_done": torch.cuda.Event(), "n": 0, "used": False}
                slot["fb"] = FrameBuffers(*[slot["dev"][n].data_ptr() for n in self.names])
                self.slots.append(slot)
            # three priority levels for three kinds of stream: the runtime multiplexes streams onto a few hardware queues and two streams of one priority can
            # share a queue -- the copies then queue behind the batch's kernels (277 ms per batch instead of 222), or the context's line stream behind the ORB
            # stream (330).  Copy streams high, the compute stream normal, the context's line stream low (api.cpp): streams of different levels never share.
            self.s_in, self.s_out = torch.cuda.Stream(self.dev, priority=-1), torch.cuda.Stream(self.dev, priority=-1)
            self.s_cmp = torch.cuda.Stream(self.dev)

    def input_buffer(self, i):
        """The pinned staging array batch i will be uploaded from ((2*pairs_per_batch, H, W) uint8).  A producer that writes its images
        straight into it (a decoder, a file reader) and passes a leading slice of it to run() saves the host-side copy."""
        return self.slots[i & 1]["host_in"].numpy()

    def _frames(self, slot):
        n = slot["n"]
        h = {k: v.numpy() for k, v in slot["host"].items()}
        f = StereoFrames()
        counts, lcounts = h["counts"][:2 * n], h["lcounts"][:2 * n]
        kps = h["kps"][:2 * n].view(KEYPOINT_DTYPE).reshape(2 * n, self.cap)
        kls = h["kls"][:2 * n].view(KEYLINE_DTYPE).reshape(2 * n, self.lcap)
        f.N, f.Nr = counts[0::2].copy(), counts[1::2].copy()
        f.mvKeys, f.mvKeysRight = kps[0::2], kps[1::2]
        f.mDescriptors, f.mDescriptorsRight = h["desc"][:2 * n][0::2], h["desc"][:2 * n][1::2]
        f.mvuRight, f.mvDepth = h["ur"][:n], h["dp"][:n]
        f.N_l, f.Nr_l = lcounts[0::2].copy(), lcounts[1::2].copy()
        f.mvKeys_Line, f.mvKeysRight_Line = kls[0::2], kls[1::2]
        f.mDescriptors_Line, f.mDescriptorsRight_Line = h["ldesc"][:2 * n][0::2], h["ldesc"][:2 * n][1::2]
        f.line_matches_12, f.mvDisparity_l, f.mvle_l = h["lm"][:n], h["ldisp"][:n], h["lle"][:n]
        return f

    def run(self, batches):
        torch = self.torch
        prev = None
        with torch.cuda.device(self.dev):
            for i, batch in enumerate(batches):
                batch = np.ascontiguousarray(batch)
                if batch.dtype != np.uint8 or batch.ndim != 3 or batch.shape[0] % 2 or batch.shape[1:] != (self.height, self.width):
                    raise TypeError("run: batches of (2*n_pairs, H, W) uint8 images of the pipeline's size expected")
                n = batch.shape[0] // 2
                if n > self.B:
                    raise ValueError("batch larger than pairs_per_batch")
                slot = self.slots[i & 1]
                if slot["used"]:
                    slot["in_done"].synchronize()                      # the staging buffer's previous upload has left the host
                if batch.ctypes.data != slot["host_in"].data_ptr():
                    slot["host_in"][:2 * n].copy_(torch.from_numpy(batch))      # (skipped when the batch already is input_buffer(i))
                slot["n"], slot["used"] = n, True
                with torch.cuda.stream(self.s_in):
                    self.s_in.wait_event(slot["cmp_done"]) if i >= 2 else None   # the path of batch i-2 has finished reading dev_in
                    slot["dev_in"][:2 * n].copy_(slot["host_in"][:2 * n], non_blocking=True)
                    slot["in_done"].record(self.s_in)
                with torch.cuda.stream(self.s_cmp):
                    self.s_cmp.wait_event(slot["in_done"])
                    if i >= 2:
                        self.s_cmp.wait_event(slot["out_done"])         # the results of batch i-2 have left the device buffers
                    if self.input_event:
                        self.ctx.set_input_event(slot["in_done"])         # the line stream waits for the upload only, not for the previous batch's tail
                    check(lib().olf_stereo_frames_dev(self.ctx.handle, C.c_void_p(slot["dev_in"].data_ptr()), n, C.byref(slot["fb"]),
                                                      C.c_void_p(self.s_cmp.cuda_stream)), "olf_stereo_frames_dev")
                    slot["cmp_done"].record(self.s_cmp)
                with torch.cuda.stream(self.s_out):
                    self.s_out.wait_event(slot["cmp_done"])
                    for k in self.names:
                        cnt = 2 * n if slot["dev"][k].shape[0] == 2 * self.B else n
                        slot["host"][k][:cnt].copy_(slot["dev"][k][:cnt], non_blocking=True)
                    slot["out_done"].record(self.s_out)
                if prev is not None:
                    prev["out_done"].synchronize()
                    self.ctx.poll_status()                                 # a batch that overflowed a device buffer is an error, not a short result
                    yield self._frames(prev)
                prev = slot
            if prev is not None:
                prev["out_done"].synchronize()
                self.ctx.poll_status()
                yield self._frames(prev)


def pcie_inclusive_rate(params, width, height, pairs=1024, batches=10, producer="pinned"):
    """bench.py leg: stereo frames/s host-to-host through the double-buffered pipeline -- host images in, every output array back in host memory at
    its fixed capacity.  `producer`: "pinned" = the images of a batch are written straight into the pipeline's pinned staging buffer, which is what a
    decoder / file reader does with `input_buffer(i)` (the buffers are filled before the clock starts: decoding is the reader's cost, not the path's);
    "pageable" = every batch is first copied from an ordinary numpy array into the staging buffer by the calling thread (the convenience shape of run()).
    `value` is the steady-state rate: the median interval between the completions of consecutive batches, i.e. without the fill (first upload, nothing to
    overlap with) and the drain (last download); `whole_run` includes both."""
    import time
    import torch
    from . import synth
    pipe = OfflinePipeline(params, width, height, pairs)
    nd = min(pairs, 64)
    base = synth.stereo_batch(7000, nd, width, height)
    batch = np.tile(base, (pairs // nd + 1, 1, 1))[:2 * pairs].copy()
    if producer == "pinned":
        for i in range(2):
            pipe.input_buffer(i)[:2 * pairs] = batch
        feed = lambda k: (pipe.input_buffer(i)[:2 * pairs] for i in range(k))
    else:
        feed = lambda k: (batch for _ in range(k))
    for _ in pipe.run(feed(2)):      # warm-up
        pass
    torch.cuda.synchronize()
    t0, n, stamps = time.perf_counter(), 0, []
    for f in pipe.run(feed(batches)):
        n += len(f.N)
        stamps.append(time.perf_counter())
    dt = time.perf_counter() - t0
    pipe.ctx.close()
    gaps = np.diff(np.array(stamps))[: max(len(stamps) - 2, 1)] if len(stamps) > 2 else np.array([dt / max(batches, 1)])      # (the last gap is the drain)
    steady = pairs / float(np.median(gaps))
    # (`value` has been the steady-state rate since round 4 -- rounds 1-3 reported the whole run under that key; both are spelled out so that rounds compare like with like)
    return {"value": round(steady, 1), "unit": "stereo frames/s", "pairs_per_batch": pairs, "batches": batches, "producer": producer,
            "steady_state": round(steady, 1), "whole_run": round(n / dt, 1), "value_is": "steady_state", "batch_interval_ms": {"median": round(float(np.median(gaps)) * 1e3, 2), "max": round(float(gaps.max()) * 1e3, 2)},
            "note": "host images in, host results out, upload / path / download overlapped on three streams; value = steady state (median interval between "
                    "batch completions), whole_run includes the pipeline's fill and drain"}
