"""Region-growing workload of the bench scene from the CPU oracle: regions by final size, and the number of <= 8-entry FIFO batches (= iterations of the
one-wave agent k_lsd_grow) they take.  python tools/grow_region_model.py [scene] [images]"""
import ctypes as C, os, sys
import numpy as np
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracle_lib as oracle
from orb_line_slam_amd import synth
scene = sys.argv[1] if len(sys.argv) > 1 else "default"
nimg = int(sys.argv[2]) if len(sys.argv) > 2 else 4
W, H = 1242, 375
p = oracle.full_params(2000, 500)
L = oracle._L
L.orc_lsd_region_stats.restype = C.c_int
imgs = synth.stereo_batch(7000, (nimg + 1) // 2, W, H, scene=scene)
tot = None
edges = [1, 2, 3, 4, 6, 8, 12, 17, 32, 64, 128, 1 << 30]
acc = np.zeros((len(edges) - 1, 10))
for k in range(nimg):
    buf = np.zeros(9 * 200000, np.int32)
    n = L.orc_lsd_region_stats(imgs[k].ctypes.data_as(C.c_void_p), W, H, C.byref(p.line), buf.ctypes.data_as(C.c_void_p), buf.size)
    t = buf[:n].reshape(-1, 9)
    for b in range(len(edges) - 1):
        m = (t[:, 0] >= edges[b]) & (t[:, 0] < edges[b + 1])
        acc[b] += [m.sum()] + [t[m, j].sum() for j in range(9)]
acc /= nimg
print(f"scene {scene}, {nimg} images, per image: regions {acc[:,0].sum():.0f}, pixels {acc[:,1].sum():.0f}, agent iterations {acc[:,2].sum():.0f}")
print(" final size    regions   pixels   iterations  it/region  size after first entry (mean) | general iterations left / entries handled by a window phase of radius 1, 2, 3")
for b in range(len(edges) - 1):
    r = max(acc[b, 0], 1e-9)
    print(f" {edges[b]:4d}-{min(edges[b+1]-1, 99999):<6d} {acc[b,0]:8.0f} {acc[b,1]:8.0f} {acc[b,2]:10.0f} {acc[b,2]/r:9.2f} {acc[b,3]/r:9.2f}      | " + "  ".join(f"{acc[b,4+2*d]:7.0f}/{acc[b,5+2*d]:7.0f}" for d in range(3)))
print("totals: " + "  ".join(f"{acc[:,4+2*d].sum():7.0f}/{acc[:,5+2*d].sum():7.0f}" for d in range(3)))
