import sys, ctypes as C, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib
n = 64
os.environ['OLF_LSD_NW'] = '0'
imgs = synth.stereo_batch(7000, 16, 1242, 375)
imgs = np.tile(imgs, (n // 32 + 1, 1, 1))[:n].copy()
if len(sys.argv) > 1 and sys.argv[1] == "bars":
    imgs = np.tile(synth.stereo_batch(7000, 16, 1242, 375, scene="bars"), (n // 32 + 1, 1, 1))[:n].copy()
if len(sys.argv) > 1 and sys.argv[1] == "tri":      # 40-pixel bands of one gradient direction: regions of thousands of pixels
    y, x = np.mgrid[0:375, 0:1242]
    imgs[:] = np.clip(np.abs((x % 48) - 24) * 10, 0, 255).astype(np.uint8)
kw = {}
if len(sys.argv) > 1 and sys.argv[1] == "wide":     # 1.5 grey levels / pixel with a low gradient threshold: a growth front of several hundred pixels
    y, x = np.mgrid[0:375, 0:1242]
    imgs[:] = np.clip(np.abs((x % 170) - 85) * 3, 0, 255).astype(np.uint8)
    kw = dict(lsd_quant=0.3, lsd_scale=2.0)
ex = ola.Lineextractor(500, 0.025, max_images=n, **kw)
k, d, c = ex.extract_batch(imgs)
out = np.zeros(128, np.int32)
_lib.lib().olf_debug_status_n(ex._ctx.handle, out.ctypes.data_as(C.c_void_p), 128)
t = out[16:76].view(np.int64)
print(dict(zip(["rounds", "k_speculated", "t_committed", "rounds_fully_committed", "single_steps", "rounds_big", "k_big", "t_big", "flushes", "iterations", "iters_fifo>=14", "iters_fifo>=21", "candidates", "grown_regions", "iters_fifo_from_memory", "nkeys", "windows_entered", "seed_lanes_gathered", "regather_lanes", "pixels_in_regions", "iso_seeds", "pixels_logged", "windows_with_live_seeds", "region_starts(first_steps)", "candidates_in_first_steps", "window_accepts", "window_entries", "window_handovers", "regions_done_in_window", "pending_at_handover"], t.tolist())))
