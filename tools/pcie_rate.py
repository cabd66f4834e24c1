"""PCIe-inclusive rate of the feature path: host images in, host results out, through orb_line_slam_amd.pipeline.OfflinePipeline.
python tools/pcie_rate.py [pairs_per_batch] [batches]"""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from orb_line_slam_amd import synth, _lib
from orb_line_slam_amd.pipeline import OfflinePipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w, h = 1242, 375
p = _lib.default_params()
pipe = OfflinePipeline(p, w, h, B)
base = synth.stereo_batch(7000, 32, w, h)
batch = np.tile(base, (B // 32 + 1, 1, 1))[:2 * B].copy()
for _ in pipe.run([batch, batch]):      # warm-up
    pass
torch.cuda.synchronize()
for mode in ("pageable input (copied into the pinned staging buffer)", "input produced in the pinned staging buffer"):
    if mode.startswith("input produced"):
        for i in range(2):
            pipe.input_buffer(i)[:] = batch
        src = (pipe.input_buffer(i) for i in range(K))
    else:
        src = (batch for _ in range(K))
    torch.cuda.synchronize()
    t = time.time(); n = 0
    for f in pipe.run(src):
        n += len(f.N)
    dt = time.time() - t
    print(f"{B} pairs per batch, {K} batches, {mode}: {n / dt:.0f} stereo frames/s host-to-host ({dt / K * 1e3:.1f} ms per batch; "
          f"mean key points {f.N.mean():.0f}, lines {f.N_l.mean():.0f})")
