"""The trimmed wire record of a batch of stereo frames: host mirror of csrc/records.hip (olf_frames_pack_dev).

What a rank sends to rank 0 in the multi-GPU mode: a header, the per-image counts and, section by section, only the rows in
use of the fixed-capacity arrays the fused entry writes (the members of a reference Frame: mvKeys / mDescriptors / mvuRight / mvDepth /
mvKeys_Line / mDescriptors_Line / stereo line matches / mvDisparity_l / mvle_l, reference include/Frame.h:180-214).  pack_records produces
the same bytes as the device kernel, parse_records splits a record back into arrays, merge_records joins the records of consecutive shards
into the record of the whole batch (what the verification of bench.py --verify and the gloo test compare).
"""
import numpy as np

MAGIC = 0x52464C4F   # 'OLFR'
# (name, row bytes, one row set per pair (rows = the LEFT image's count) instead of per image, counted by lcounts instead of counts)
SECTIONS = (("kps", 28, False, False), ("desc", 32, False, False), ("uright", 4, True, False), ("depth", 4, True, False),
            ("kls", 68, False, True), ("ldesc", 32, False, True), ("lmatches12", 4, True, True), ("ldisp", 8, True, True), ("lle", 24, True, True))


def _align16(x):
    return (x + 15) & ~15


def _rows(arr, n_sets, row_bytes):
    a = np.ascontiguousarray(arr)
    if n_sets == 0:      # an empty shard (more ranks than frames): no rows, the capacity still comes from the array's second dimension
        return np.zeros((0, a.shape[1] if a.ndim >= 2 else 0, row_bytes), np.uint8)
    return a.view(np.uint8).reshape(n_sets, -1, row_bytes)


def pack_records(arrays, counts, lcounts):
    """arrays: dict name -> full-capacity array ([2n] or [n] leading dimension as in olf_frame_buffers); counts / lcounts: int32 [2n]."""
    counts = np.asarray(counts, np.int32); lcounts = np.asarray(lcounts, np.int32)
    n2 = len(counts); n = n2 // 2
    cap = _rows(arrays["kps"], n2, 28).shape[1]
    lcap = _rows(arrays["kls"], n2, 68).shape[1]
    tot = (int(counts.sum()), int(lcounts.sum()), int(counts[0::2].sum()), int(lcounts[0::2].sum()))
    hdr = np.zeros(16, np.uint32)
    hdr[:8] = (MAGIC, n, cap, lcap) + tot
    parts = [hdr.tobytes(), counts.tobytes(), lcounts.tobytes()]
    size = 64 + 8 * n2
    pad = _align16(size) - size
    parts.append(b"\0" * pad); size += pad
    for name, rb, per_pair, line in SECTIONS:
        cnt = lcounts if line else counts
        rows = _rows(arrays[name], n if per_pair else n2, rb)
        sel = [rows[i, :cnt[2 * i if per_pair else i]] for i in range(n if per_pair else n2)]
        blob = (np.concatenate(sel) if sel else np.zeros((0, rb), np.uint8)).tobytes()
        pad = _align16(len(blob)) - len(blob)
        parts += [blob, b"\0" * pad]
        size += len(blob) + pad
    out = b"".join(parts)
    assert len(out) == size
    return out


def parse_records(buf):
    """bytes-like -> dict: n_pairs, cap, lcap, counts, lcounts and per section the concatenated rows as uint8 [rows, row bytes]."""
    b = np.frombuffer(bytes(buf), np.uint8)
    hdr = b[:64].view(np.uint32)
    if hdr[0] != MAGIC:
        raise ValueError("not a frame record")
    n = int(hdr[1]); n2 = 2 * n
    out = {"n_pairs": n, "cap": int(hdr[2]), "lcap": int(hdr[3])}
    out["counts"] = b[64:64 + 4 * n2].view(np.int32).copy()
    out["lcounts"] = b[64 + 4 * n2:64 + 8 * n2].view(np.int32).copy()
    tot = {(False, False): int(hdr[4]), (False, True): int(hdr[5]), (True, False): int(hdr[6]), (True, True): int(hdr[7])}
    assert tot[(False, False)] == out["counts"].sum() and tot[(False, True)] == out["lcounts"].sum()
    off = _align16(64 + 8 * n2)
    for name, rb, per_pair, line in SECTIONS:
        rows = tot[(per_pair, line)]
        out[name] = b[off:off + rows * rb].reshape(rows, rb).copy()
        off = _align16(off + rows * rb)
    out["bytes"] = off
    return out


def merge_records(records):
    """Records of consecutive shards (bytes-like each) -> the record of the concatenated batch (capacities must agree)."""
    ps = [parse_records(r) for r in records]
    if not ps:
        raise ValueError("no records")
    assert len({(p["cap"], p["lcap"]) for p in ps}) == 1, "records of contexts with different capacities"
    counts = np.concatenate([p["counts"] for p in ps]); lcounts = np.concatenate([p["lcounts"] for p in ps])
    n2 = len(counts)
    hdr = np.zeros(16, np.uint32)
    hdr[:8] = (MAGIC, n2 // 2, ps[0]["cap"], ps[0]["lcap"], counts.sum(), lcounts.sum(), counts[0::2].sum(), lcounts[0::2].sum())
    parts = [hdr.tobytes(), counts.tobytes(), lcounts.tobytes()]
    size = 64 + 8 * n2
    parts.append(b"\0" * (_align16(size) - size))
    for name, rb, _, _ in SECTIONS:
        blob = np.concatenate([p[name] for p in ps]).tobytes()
        parts += [blob, b"\0" * (_align16(len(blob)) - len(blob))]
    return b"".join(parts)
