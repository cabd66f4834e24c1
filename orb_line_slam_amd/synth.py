"""Seeded integer-only synthetic stereo images (csrc/synth.c).

The reference has no images in its tree; its examples read KITTI/EuRoC from a user path
(Examples/PL/PL_stereo_kitti.cc:47-58).  Tests and bench.py use these instead; the generator is
pure integer arithmetic so every machine produces the same bytes for a given (seed, W, H).
"""
import ctypes as C
import os
import numpy as np
from ._lib import SYNTH_PATH

_syn = None


def _lib():
    global _syn
    if _syn is None:
        if not os.path.exists(SYNTH_PATH):
            raise ImportError(f"{SYNTH_PATH} is missing: run `make -C orb_line_slam_amd/csrc`")
        _syn = C.CDLL(SYNTH_PATH)
        _syn.olf_synth_stereo.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _syn.olf_synth_stereo_scene.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return _syn


SCENES = {"default": 0, "long": 1, "bars": 2}     # long: fewer, larger shapes -> segments of about 0.08 * W pixels (SURVEY App. D's model of a KITTI frame)


def stereo_pair(seed, width, height, scene="default"):
    """(left, right) uint8 arrays of shape (height, width)."""
    left = np.empty((height, width), np.uint8)
    right = np.empty((height, width), np.uint8)
    rc = _lib().olf_synth_stereo_scene(int(seed), int(width), int(height), SCENES[scene], left.ctypes.data_as(C.c_void_p),
                                       right.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError("olf_synth_stereo: bad size")
    return left, right


def stereo_batch(base_seed, n_pairs, width, height, scene="default"):
    """uint8 array (2*n_pairs, height, width): image 2*i = left of pair i, 2*i+1 = right."""
    out = np.empty((2 * n_pairs, height, width), np.uint8)
    for i in range(n_pairs):
        l, r = stereo_pair(base_seed + i, width, height, scene)
        out[2 * i], out[2 * i + 1] = l, r
    return out
