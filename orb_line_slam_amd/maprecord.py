"""The per-key-frame feature record of the binary map file (Map::SaveKeyFrame src/Map.cc:283-373, Map::LoadKeyFrame :376-531;
SURVEY.md 8(f) rank 4): a byte-exact serialisation of exactly the path's outputs -- every key point (6 x 4 B), uRight, depth, 32-byte ORB
descriptor, MapPoint id; every KeyLine (17 fields), disparities, le_l (3 doubles), 32-byte LBD descriptor, MapLine id.  The packing and
unpacking are host code inside the library (csrc/maprecord.cpp: olf_kf_record_pack / _counts / _unpack); this module passes the arrays
StereoFrames.pair() returns.
"""
import ctypes as C
import numpy as np
from ._lib import KEYLINE_DTYPE, KEYPOINT_DTYPE, check, lib

ULONG_MAX = np.uint64(0xFFFFFFFFFFFFFFFF)        # "no MapPoint / MapLine" (src/Map.cc:318-322)


def _api():
    L = lib()
    if not getattr(L, "_kf_record_ready", False):
        L.olf_kf_record_bytes.restype = C.c_size_t
        L.olf_kf_record_bytes.argtypes = [C.c_int, C.c_int]
        L.olf_kf_record_pack.argtypes = [C.c_uint64, C.c_uint64, C.c_double, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 5 + \
                                        [C.c_int] + [C.c_void_p] * 5 + [C.c_void_p, C.c_size_t, C.c_void_p]
        L.olf_kf_record_counts.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.olf_kf_record_unpack.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 15
        L._kf_record_ready = True
    return L


def _arr(a, dt, shape=None):
    b = np.ascontiguousarray(np.asarray(a, dt))
    return b if shape is None else b.reshape(shape)


def pack_keyframe(mnFrameId, mnId, mTimeStamp, t, quat, mvKeys, mvuRight, mvDepth, mDescriptors, mappoint_ids=None, mvKeys_Line=None,
                  mvDisparity_l=None, mvle_l=None, mDescriptors_l=None, mapline_ids=None):
    """-> bytes written by Map::SaveKeyFrame (built with HasLine) for one key frame.  *_ids: uint64 arrays, ULONG_MAX where NULL."""
    L = _api()
    n = len(mvKeys)
    nl = 0 if mvKeys_Line is None else len(mvKeys_Line)
    keep = [_arr(t, np.float32), _arr(quat, np.float32), _arr(mvKeys, KEYPOINT_DTYPE), _arr(mvuRight, np.float32), _arr(mvDepth, np.float32),
            _arr(mDescriptors, np.uint8, (n, 32)), None if mappoint_ids is None else _arr(mappoint_ids, np.uint64)]
    if nl:
        keep += [_arr(mvKeys_Line, KEYLINE_DTYPE), _arr(mvDisparity_l, np.float32, (nl, 2)), _arr(mvle_l, np.float64, (nl, 3)),
                 _arr(mDescriptors_l, np.uint8, (nl, 32)), None if mapline_ids is None else _arr(mapline_ids, np.uint64)]
    else:
        keep += [None] * 5
    p = [None if a is None or a.size == 0 else a.ctypes.data for a in keep]
    out = np.zeros(L.olf_kf_record_bytes(n, nl), np.uint8)
    written = C.c_size_t(0)
    check(L.olf_kf_record_pack(int(mnFrameId), int(mnId), float(mTimeStamp), p[0], p[1], n, p[2], p[3], p[4], p[5], p[6], nl, p[7], p[8], p[9],
                               p[10], p[11], out.ctypes.data, out.size, C.byref(written)), "olf_kf_record_pack")
    return out[:written.value].tobytes()


def unpack_keyframe(buf, offset=0):
    """Map::LoadKeyFrame's reads. -> (dict, next offset)"""
    L = _api()
    raw = np.frombuffer(buf, np.uint8, offset=offset)
    n, nl, total = C.c_int32(0), C.c_int32(0), C.c_size_t(0)
    check(L.olf_kf_record_counts(raw.ctypes.data, raw.size, C.byref(n), C.byref(nl), C.byref(total)), "olf_kf_record_counts")
    n, nl = n.value, nl.value
    fid, kid, ts = C.c_uint64(0), C.c_uint64(0), C.c_double(0)
    t, q = np.zeros(3, np.float32), np.zeros(4, np.float32)
    keys, ur, dp = np.zeros(n, KEYPOINT_DTYPE), np.zeros(n, np.float32), np.zeros(n, np.float32)
    desc, mp = np.zeros((n, 32), np.uint8), np.zeros(n, np.uint64)
    lines, disp, le = np.zeros(nl, KEYLINE_DTYPE), np.zeros((nl, 2), np.float32), np.zeros((nl, 3), np.float64)
    ldesc, ml = np.zeros((nl, 32), np.uint8), np.zeros(nl, np.uint64)
    check(L.olf_kf_record_unpack(raw.ctypes.data, raw.size, C.byref(fid), C.byref(kid), C.byref(ts), t.ctypes.data, q.ctypes.data, keys.ctypes.data,
                                 ur.ctypes.data, dp.ctypes.data, desc.ctypes.data, mp.ctypes.data, lines.ctypes.data, disp.ctypes.data,
                                 le.ctypes.data, ldesc.ctypes.data, ml.ctypes.data), "olf_kf_record_unpack")
    return dict(mnFrameId=fid.value, mnId=kid.value, mTimeStamp=ts.value, t=t, quat=q, mvKeys=keys, mvuRight=ur, mvDepth=dp, mDescriptors=desc,
                mappoint_ids=mp, mvKeys_Line=lines, mvDisparity_l=disp, mvle_l=le, mDescriptors_l=ldesc, mapline_ids=ml), offset + total.value
