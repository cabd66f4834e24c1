"""The per-key-frame feature record of the binary map file (Map::SaveKeyFrame src/Map.cc:283-373, Map::LoadKeyFrame :376-531;
SURVEY.md 8(f) rank 4): a byte-exact serialisation of exactly the path's outputs -- every key point (6 x 4 B), uRight, depth, 32-byte ORB
descriptor, MapPoint id; every KeyLine (17 fields), disparities, le_l (3 doubles), 32-byte LBD descriptor, MapLine id.  Host code, like
the reference's; the arrays are the ones StereoFrames.pair() returns.
"""
import struct
import numpy as np
from ._lib import KEYLINE_DTYPE, KEYPOINT_DTYPE

ULONG_MAX = np.uint64(0xFFFFFFFFFFFFFFFF)        # "no MapPoint / MapLine" (src/Map.cc:318-322)

_HEADER = struct.Struct("<QQd3f4f")              # mnFrameId, mnId, mTimeStamp, t (Tcw(0..2, 3)), quaternion (Converter::toQuaternion)
KP_RECORD = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"),
                      ("uRight", "<f4"), ("depth", "<f4"), ("desc", "u1", (32,)), ("mp", "<u8")])
KL_RECORD = np.dtype([(n, KEYLINE_DTYPE.fields[n][0]) for n in KEYLINE_DTYPE.names] +
                     [("disp_first", "<f4"), ("disp_second", "<f4"), ("le", "<f8", (3,)), ("desc", "u1", (32,)), ("ml", "<u8")])
assert _HEADER.size == 52 and KP_RECORD.itemsize == 72 and KL_RECORD.itemsize == 140


def pack_keyframe(mnFrameId, mnId, mTimeStamp, t, quat, mvKeys, mvuRight, mvDepth, mDescriptors, mappoint_ids=None, mvKeys_Line=None,
                  mvDisparity_l=None, mvle_l=None, mDescriptors_l=None, mapline_ids=None):
    """-> bytes written by Map::SaveKeyFrame (built with HasLine) for one key frame.  *_ids: uint64 arrays, ULONG_MAX where NULL."""
    n = len(mvKeys)
    kp = np.zeros(n, KP_RECORD)
    for f in ("x", "y", "size", "angle", "response", "octave"):
        kp[f] = mvKeys[f]
    kp["uRight"], kp["depth"] = mvuRight, mvDepth
    kp["desc"] = np.asarray(mDescriptors, np.uint8).reshape(n, 32)
    kp["mp"] = ULONG_MAX if mappoint_ids is None else mappoint_ids
    nl = 0 if mvKeys_Line is None else len(mvKeys_Line)
    kl = np.zeros(nl, KL_RECORD)
    if nl:
        for f in KEYLINE_DTYPE.names:
            kl[f] = mvKeys_Line[f]
        d = np.asarray(mvDisparity_l, np.float32).reshape(nl, 2)
        kl["disp_first"], kl["disp_second"] = d[:, 0], d[:, 1]
        kl["le"] = np.asarray(mvle_l, np.float64).reshape(nl, 3)
        kl["desc"] = np.asarray(mDescriptors_l, np.uint8).reshape(nl, 32)
        kl["ml"] = ULONG_MAX if mapline_ids is None else mapline_ids
    head = _HEADER.pack(int(mnFrameId), int(mnId), float(mTimeStamp), *[float(np.float32(v)) for v in t], *[float(np.float32(v)) for v in quat])
    return head + struct.pack("<i", n) + kp.tobytes() + struct.pack("<i", nl) + kl.tobytes()


def unpack_keyframe(buf, offset=0):
    """Map::LoadKeyFrame's reads. -> (dict, next offset)"""
    mnFrameId, mnId, ts, *tq = _HEADER.unpack_from(buf, offset)
    offset += _HEADER.size
    (n,) = struct.unpack_from("<i", buf, offset); offset += 4
    kp = np.frombuffer(buf, KP_RECORD, n, offset); offset += n * KP_RECORD.itemsize
    (nl,) = struct.unpack_from("<i", buf, offset); offset += 4
    kl = np.frombuffer(buf, KL_RECORD, nl, offset); offset += nl * KL_RECORD.itemsize
    keys = np.zeros(n, KEYPOINT_DTYPE)
    for f in ("x", "y", "size", "angle", "response", "octave"):
        keys[f] = kp[f]
    keys["class_id"] = -1                      # not stored; cv::KeyPoint's default
    lines = np.zeros(nl, KEYLINE_DTYPE)
    for f in KEYLINE_DTYPE.names:
        lines[f] = kl[f]
    return dict(mnFrameId=mnFrameId, mnId=mnId, mTimeStamp=ts, t=np.array(tq[:3], np.float32), quat=np.array(tq[3:], np.float32), mvKeys=keys,
                mvuRight=kp["uRight"].copy(), mvDepth=kp["depth"].copy(), mDescriptors=kp["desc"].copy(), mappoint_ids=kp["mp"].copy(),
                mvKeys_Line=lines, mvDisparity_l=np.stack([kl["disp_first"], kl["disp_second"]], 1) if nl else np.zeros((0, 2), np.float32),
                mvle_l=kl["le"].copy(), mDescriptors_l=kl["desc"].copy(), mapline_ids=kl["ml"].copy()), offset
