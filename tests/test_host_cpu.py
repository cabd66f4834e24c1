"""CPU tests (no GPU): the C ABI loads and exports what include/orbline.h declares, host-side logic, the device
math routines compiled for the host, the synthetic generator, and the multi-process sharding/gather on gloo."""
import ctypes as C
import os
import re
import subprocess
import sys
import zlib
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    import orb_line_slam_amd as ola
    L = ola.lib()                                    # loads liborbline_hip.so (no GPU needed to load)
    hdr = open(os.path.join(ROOT, "include", "orbline.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(olf_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/orbline.h but not exported"


def test_no_cpu_fallback_without_gpu():
    import orb_line_slam_amd as ola
    if ola.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(ola.OlfError) as e:
        ola.ORBextractor(1000, 1.2, 8, 20, 7)(np.zeros((480, 640), np.uint8))
    assert e.value.code == -4                        # OLF_ERR_NODEVICE: the product has no CPU path
    with pytest.raises(ola.OlfError):
        ola.StereoFrontEnd(None, 640, 480, 1)


def test_params_block_must_come_from_default_params():
    """olf_params carries abi_version + struct_size, stamped by olf_default_params and checked by olf_ctx_create before anything
    else (also before the device check), so a block laid out by an older header is refused instead of being read past its end"""
    import ctypes as C
    from orb_line_slam_amd import _lib
    L = _lib.lib()
    p = _lib.default_params()
    assert p.abi_version >= 3 and p.struct_size == C.sizeof(_lib.OlfParams)
    ctx = C.c_void_p()
    q = _lib.OlfParams()                                  # zero-initialised: no stamp
    assert L.olf_ctx_create(C.byref(q), 640, 480, 1, C.byref(ctx)) == -1 and not ctx.value
    assert b"olf_default_params" in L.olf_last_error()
    p.struct_size -= 4                                    # a caller whose header lacks the last field
    assert L.olf_ctx_create(C.byref(p), 640, 480, 1, C.byref(ctx)) == -1 and not ctx.value


def test_record_layouts_match_the_reference_types():
    import orb_line_slam_amd as ola
    assert ola.KEYPOINT_DTYPE.itemsize == 28 and ola.KEYLINE_DTYPE.itemsize == 68
    assert ola.KEYPOINT_DTYPE.names == ("x", "y", "size", "angle", "response", "octave", "class_id")
    assert ola.KEYLINE_DTYPE.names[:3] == ("angle", "class_id", "octave") and ola.KEYLINE_DTYPE.names[-1] == "numOfPixels"
    p = ola.default_params()
    assert (p.orb.nfeatures, p.orb.nlevels, p.orb.ini_th_fast, p.orb.min_th_fast) == (2000, 8, 20, 7)
    assert (p.line.lsd_nfeatures, p.line.lsd_n_bins, p.stereo.matching_s_ws) == (500, 1024, 10)
    assert abs(p.stereo.bf - 386.1448) < 1e-4 and p.stereo.best_lr_matches == 1
    assert ola.ORBmatcher.TH_LOW == 50 and ola.ORBmatcher.TH_HIGH == 100 and ola.ORBmatcher.HISTO_LENGTH == 30


def test_input_validation_mirrors_reference_errors():
    import orb_line_slam_amd as ola
    with pytest.raises(TypeError):                   # assert(image.type() == CV_8UC1), src/ORBextractor.cc:1052
        ola.ORBextractor(1000, 1.2, 8, 20, 7).extract_batch(np.zeros((1, 480, 640), np.float32))
    with pytest.raises(RuntimeError):                # "Error, depth image!= 0", LSDDetector_custom.cpp:236-237
        ola.Lineextractor(200, 0.025).extract_batch(np.zeros((1, 480, 640), np.float32))
    k, d = ola.Lineextractor(200, 0.025, bFLD=True)(np.zeros((480, 640), np.uint8))   # bFLD: returns nothing (src/LineExtractor.cc:68)
    assert len(k) == 0 and d.shape == (0, 32)


@pytest.fixture(scope="module")
def hostmath(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("hm") / "libhostmath.so")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "hostmath.cpp")], check=True)
    L = C.CDLL(so)
    L.hm_cosf.restype = L.hm_sinf.restype = L.hm_fast_atan2.restype = C.c_float
    L.hm_cosf.argtypes = L.hm_sinf.argtypes = [C.c_float]
    L.hm_fast_atan2.argtypes = [C.c_float, C.c_float]
    L.hm_sweep_sincos.restype = C.c_long
    L.hm_sweep_sincos.argtypes = [C.c_float, C.c_float, C.c_uint]
    return L


def test_device_sincosf_equals_libm(hostmath):
    # the descriptor stage evaluates cosf/sinf(angle * pi/180), angle in [0, 360): every 7th float of [0, 6.3]
    # (the full 1.09e9-value sweep is bit-identical too; it takes ~4 s and is run by hand)
    assert hostmath.hm_sweep_sincos(0.0, 6.3, 7) == 0
    assert hostmath.hm_cosf(0.0) == 1.0 and hostmath.hm_sinf(0.0) == 0.0


def test_device_atan2f_and_signed_sincosf_equal_libm(hostmath):
    """Convention C.6, variant conv_libm_float = 1: KeyLine.angle = atan2f(dy, dx) and LBD's cosf / sinf(direction), direction in [-pi, pi].  The device
    routines (device_math.hpp, compiled here for the host from the same source) against this box's glibc: 20 million pseudo-random float pairs + a grid for
    atan2f (200 million are bit-identical too, run by hand), every 5th negative float of [-3.2, 0] for cosf / sinf."""
    import ctypes as C
    hostmath.hm_sweep_atan2f.restype = C.c_long; hostmath.hm_sweep_atan2f.argtypes = [C.c_ulong, C.c_long]
    hostmath.hm_sweep_sincos_neg.restype = C.c_long; hostmath.hm_sweep_sincos_neg.argtypes = [C.c_float, C.c_float, C.c_uint]
    assert hostmath.hm_sweep_atan2f(2026, 20_000_000) == 0
    assert hostmath.hm_sweep_sincos_neg(0.0, 3.2, 5) == 0


def test_device_fast_atan2_equals_oracle(hostmath, oracle):
    rng = np.random.default_rng(7)
    ys = np.concatenate([rng.integers(-200000, 200000, 4000).astype(np.float32), rng.normal(0, 1, 2000).astype(np.float32), [0, 0, 1, -1]])
    xs = np.concatenate([rng.integers(-200000, 200000, 4000).astype(np.float32), rng.normal(0, 1, 2000).astype(np.float32), [0, 5, 0, 0]])
    for y, x in zip(ys, xs):
        a, b = np.float32(hostmath.hm_fast_atan2(float(y), float(x))), np.float32(oracle.fast_atan2(float(y), float(x)))
        assert a.view(np.uint32) == b.view(np.uint32), (y, x, a, b)


def test_synth_is_deterministic_and_has_disparity():
    from orb_line_slam_amd import synth
    l1, r1 = synth.stereo_pair(7, 640, 480)
    l2, r2 = synth.stereo_pair(7, 640, 480)
    assert np.array_equal(l1, l2) and np.array_equal(r1, r2)
    l3, _ = synth.stereo_pair(8, 640, 480)
    assert not np.array_equal(l1, l3)
    # right(x, y) ~ left(x + d(y), y), d(y) = 4 + 60*y/H, up to the +-3 noise of each image
    y = 240
    d = 4 + (60 * y) // 480
    diff = np.abs(l1[y, d:600].astype(int) - r1[y, :600 - d].astype(int))
    assert diff.max() <= 6
    b = synth.stereo_batch(3, 2, 640, 480)
    assert b.shape == (4, 480, 640) and np.array_equal(b[2], synth.stereo_pair(4, 640, 480)[0])


def test_shard_range_tiles_the_frames():
    from orb_line_slam_amd.distributed import shard_range
    for n in (0, 1, 7, 8, 64, 1000):
        for world in (1, 2, 3, 4, 8):
            blocks = [shard_range(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from orb_line_slam_amd.distributed import shard_range, gather_records, SizeExchange
from orb_line_slam_amd.records import pack_records, merge_records, parse_records
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=rank, world_size=world)
# full-capacity output arrays of 7 stereo frames, as the fused entry writes them: rows from the committed fixture of the feature path
# (tests/golden/frame_320x240_seed11.npz), a different number of rows in use per frame
g = np.load(os.path.join(sys.argv[1], "tests", "golden", "frame_320x240_seed11.npz"))
n_frames, cap, lcap = (int(sys.argv[3]) if len(sys.argv) > 3 else 7), 520, 100
def frames(lo, hi):
    n = hi - lo
    a = {"kps": np.zeros((2 * n, cap, 28), np.uint8), "desc": np.zeros((2 * n, cap, 32), np.uint8), "uright": np.full((n, cap), -1, np.float32),
         "depth": np.full((n, cap), -1, np.float32), "kls": np.zeros((2 * n, lcap, 68), np.uint8), "ldesc": np.zeros((2 * n, lcap, 32), np.uint8),
         "lmatches12": np.full((n, lcap), -1, np.int32), "ldisp": np.full((n, lcap, 2), -1, np.float32), "lle": np.zeros((n, lcap, 3), np.float64)}
    counts, lcounts = np.zeros(2 * n, np.int32), np.zeros(2 * n, np.int32)
    for q, f in enumerate(range(lo, hi)):
        nl_, nr_ = 504 - 31 * (f % 8), 300 + 7 * (f % 8)
        ll_, lr_ = 100 - 9 * (f % 8), 40 + 3 * (f % 8)
        counts[2 * q], counts[2 * q + 1], lcounts[2 * q], lcounts[2 * q + 1] = nl_, nr_, ll_, lr_
        kb, klb = g["kpsL"].view(np.uint8).reshape(-1, 28), g["klsL"].view(np.uint8).reshape(-1, 68)
        a["kps"][2 * q, :nl_] = kb[:nl_]; a["kps"][2 * q + 1, :nr_] = np.roll(kb, f, 0)[:nr_]
        a["desc"][2 * q, :nl_] = g["descL"][:nl_]; a["desc"][2 * q + 1, :nr_] = np.roll(g["descL"], f, 0)[:nr_]
        a["uright"][q, :nl_] = g["uRight"][:nl_]; a["depth"][q, :nl_] = g["depth"][:nl_]
        a["kls"][2 * q, :ll_] = klb[:ll_]; a["kls"][2 * q + 1, :lr_] = np.roll(klb, f, 0)[:lr_]
        a["ldesc"][2 * q, :ll_] = g["ldescL"][:ll_]; a["ldesc"][2 * q + 1, :lr_] = np.roll(g["ldescL"], f, 0)[:lr_]
        a["lmatches12"][q, :ll_] = g["lm12"][:ll_]; a["ldisp"][q, :ll_] = g["ldisp"][:ll_]; a["lle"][q, :ll_] = g["lle"][:ll_]
    return a, counts, lcounts
lo, hi = shard_range(n_frames, rank, world)
rec = pack_records(*frames(lo, hi))
packed = torch.from_numpy(np.frombuffer(rec + b"\\0" * 100, np.uint8).copy())          # the device buffer is larger than the record
recs, sizes = gather_records(packed, len(rec), dist)
assert sizes[rank] == len(rec)
# the same gather with the sizes exchanged ahead of it (SizeExchange: what the pipelined bench does -- no size exchange inside the transfer)
sx = SizeExchange(torch.tensor([len(rec)], dtype=torch.int64), dist)
recs2, sizes2 = gather_records(packed, None, dist, sizes=sx.sizes())
assert sizes2 == sizes and (recs2 is None) == (recs is None)
if recs is not None:
    assert all(a.numpy().tobytes() == b.numpy().tobytes() for a, b in zip(recs, recs2))
if rank == 0:
    whole = pack_records(*frames(0, n_frames))                                          # what a 1-rank run of the same frames packs
    got = merge_records([r.numpy().tobytes() for r in recs])
    assert got == whole, "gathered records differ from the single-rank record"
    p = parse_records(got)
    assert p["n_pairs"] == n_frames and p["counts"][0] == 504 and len(p["kps"]) == p["counts"].sum() and p["bytes"] == len(whole)
    assert all(sizes[r] > 0 for r in range(world)), sizes            # (an empty shard still sends its header)
    print("GATHER_OK", len(whole), sizes)
else:
    assert recs is None
dist.barrier(); dist.destroy_process_group()
'''


def test_frame_sharded_gather_world2_gloo(tmp_path):
    """N>1 path on CPU: 2 processes, gloo, 7 frames sharded 4+3, each rank packs the trimmed record of its frames (fixture rows) and
    sends it to rank 0; the merged records must be byte-identical to the record of a 1-rank run (SURVEY 8(e) "Verification")."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(port)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK" in outs[0]


@pytest.mark.parametrize("n_frames", [7, 13])
def test_frame_sharded_gather_world8_gloo(tmp_path, n_frames):
    """The shape of the driver's 8-GPU run, on CPU: 8 gloo ranks; 7 frames leave rank 7 EMPTY (a zero-pair record still takes part in the size
    exchange and the point-to-point gather), 13 frames give uneven shards (2, 2, 2, 2, 2, 1, 1, 1).  Rank 0's merge must equal the 1-rank record."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(8):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(port), str(n_frames)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK" in outs[0]


def test_cpp_adaptor_compiles_and_links(tmp_path):
    """include/orbline_adaptor.hpp (the reference class surfaces over the C ABI) builds with plain g++ and links the library."""
    exe = str(tmp_path / "adaptor_check")
    libdir = os.path.join(ROOT, "orb_line_slam_amd", "csrc")
    subprocess.run(["g++", "-std=c++11", "-O1", "-o", exe, os.path.join(ROOT, "tests", "adaptor_compile.cpp"), "-L" + libdir, "-lorbline_hip",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert out.returncode == 0 and b"ADAPTOR_OK" in out.stdout, out.stdout


def _build_reference_api(tmp_path):
    exe = str(tmp_path / "reference_api_check")
    libdir = os.path.join(ROOT, "orb_line_slam_amd", "csrc")
    subprocess.run(["g++", "-std=c++11", "-O1", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "adaptor_reference_api.cpp"), "-L" + libdir,
                    "-lorbline_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_reference_signatures_instantiate(tmp_path):
    """include/orbline_reference_api.hpp: every template with the reference's own signature (all of include/ORBmatcher.h:37-103 -- SearchByProjection
    x5, SearchByBoW x2, SearchForInitialization, SearchForTriangulation, Fuse x2, SearchBySim3, DescriptorDistance -- match x2, matchNNR, distance,
    matchGrid x2, GridStructure, getLineCoords, StereoFrameFeatures) instantiates with
    stand-ins that carry the reference's member names, links, and -- without a device -- throws instead of falling back."""
    out = subprocess.run([_build_reference_api(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert out.returncode == 0 and b"REFERENCE_API_COMPILED" in out.stdout, out.stdout


def test_keyframe_record_bytes(oracle):
    """Map::SaveKeyFrame's byte layout: the host packer equals the oracle's field-by-field writes; unpack inverts it"""
    import ctypes as C
    from orb_line_slam_amd import maprecord, _lib
    rng = np.random.default_rng(4)
    n, nl = 37, 11
    keys = np.zeros(n, _lib.KEYPOINT_DTYPE)
    for f in ("x", "y", "size", "angle", "response"):
        keys[f] = rng.random(n, np.float32) * 100
    keys["octave"] = rng.integers(0, 8, n); keys["class_id"] = -1
    kls = np.zeros(nl, _lib.KEYLINE_DTYPE)
    for f in _lib.KEYLINE_DTYPE.names:
        kls[f] = rng.integers(0, 500, nl) if kls.dtype[f].kind == "i" else rng.random(nl, np.float32) * 300
    ur, dp = rng.random(n, np.float32) * 50 - 1, rng.random(n, np.float32) * 30
    desc, dl = rng.integers(0, 256, (n, 32), dtype=np.uint8), rng.integers(0, 256, (nl, 32), dtype=np.uint8)
    mp = rng.integers(0, 1 << 40, n).astype(np.uint64); mp[::5] = maprecord.ULONG_MAX
    ml = rng.integers(0, 1 << 40, nl).astype(np.uint64); ml[::3] = maprecord.ULONG_MAX
    disp, le = rng.random((nl, 2), np.float32) * 9, rng.standard_normal((nl, 3))
    t, q = rng.standard_normal(3).astype(np.float32), rng.standard_normal(4).astype(np.float32)
    got = maprecord.pack_keyframe(123456789012, 42, 1403636579.763555527, t, q, keys, ur, dp, desc, mp, kls, disp, le, dl, ml)
    out = np.zeros(len(got) + 64, np.uint8)
    L = oracle._L
    L.orc_keyframe_record.restype = C.c_size_t
    L.orc_keyframe_record.argtypes = [C.c_ulong, C.c_ulong, C.c_double] + [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 6
    p = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
    keep = [np.ascontiguousarray(a) for a in (t, q, keys, ur, dp, desc, mp, kls, disp, le, dl, ml)]
    size = L.orc_keyframe_record(123456789012, 42, 1403636579.763555527, p(keep[0]), p(keep[1]), n, *[p(a) for a in keep[2:7]], nl,
                                 *[p(a) for a in keep[7:]], out.ctypes.data_as(C.c_void_p))
    assert size == len(got) == 52 + 4 + 72 * n + 4 + 140 * nl
    assert out[:size].tobytes() == got
    rec, end = maprecord.unpack_keyframe(got + b"tail")
    assert end == len(got) and rec["mnFrameId"] == 123456789012 and rec["mnId"] == 42 and rec["mTimeStamp"] == 1403636579.763555527
    assert np.array_equal(rec["mvKeys"], keys) and np.array_equal(rec["mvKeys_Line"], kls) and np.array_equal(rec["mappoint_ids"], mp)
    assert np.array_equal(rec["mvle_l"], le) and np.array_equal(rec["mvDisparity_l"], disp) and np.array_equal(rec["mDescriptors_l"], dl)
    assert np.array_equal(rec["mvuRight"], ur) and np.array_equal(rec["mvDepth"], dp) and np.array_equal(rec["mDescriptors"], desc)
    empty = maprecord.pack_keyframe(0, 0, 0.0, [0, 0, 0], [0, 0, 0, 1], np.zeros(0, _lib.KEYPOINT_DTYPE), [], [], np.zeros((0, 32), np.uint8))
    assert len(empty) == 60 and maprecord.unpack_keyframe(empty)[1] == 60
    # a truncated record is reported, not read past its end (Map::LoadKeyFrame would read garbage)
    for cut in (10, 55, 60 + 71, len(got) - 1):
        with pytest.raises(_lib.OlfError):
            maprecord.unpack_keyframe(got[:cut])
    # two records back to back, as in the map file
    both = got + empty
    r1, o1 = maprecord.unpack_keyframe(both)
    r2, o2 = maprecord.unpack_keyframe(both, o1)
    assert o1 == len(got) and o2 == len(both) and len(r2["mvKeys"]) == 0 and r1["mnId"] == 42


def test_default_params_match_reference_config():
    """olf_default_params against the defaults of the reference's Config::Config() (src/Config.cpp:27-110), read from the checkout when it
    is present (build container only): every stereo-line / LSD parameter the path uses, except lsd_nfeatures (BASELINE's KITTI config)."""
    import re
    from orb_line_slam_amd import _lib
    path = "/root/reference/src/Config.cpp"
    if not os.path.exists(path):
        pytest.skip("reference checkout absent")
    text = open(path, encoding="utf-8", errors="replace").read()
    ref = {}
    for name, val in re.findall(r"^\s*(\w+)\s*=\s*([-+0-9.eE]+|true|false)\s*;", text, re.M):
        ref.setdefault(name, {"true": 1.0, "false": 0.0}.get(val, None) if val in ("true", "false") else float(val))
    p = _lib.default_params()
    got = {"min_disp": p.stereo.min_disp, "line_sim_th": p.stereo.line_sim_th, "stereo_overlap_th": p.stereo.stereo_overlap_th,
           "line_horiz_th": p.stereo.line_horiz_th, "min_ratio_12_l": p.stereo.min_ratio_12_l, "ls_min_disp_ratio": p.stereo.ls_min_disp_ratio,
           "best_lr_matches": p.stereo.best_lr_matches, "matching_s_ws": p.stereo.matching_s_ws, "min_line_length": p.line.min_line_length,
           "lsd_refine": p.line.lsd_refine, "lsd_scale": p.line.lsd_scale, "lsd_sigma_scale": p.line.lsd_sigma_scale, "lsd_quant": p.line.lsd_quant,
           "lsd_ang_th": p.line.lsd_ang_th, "lsd_log_eps": p.line.lsd_log_eps, "lsd_density_th": p.line.lsd_density_th, "lsd_n_bins": p.line.lsd_n_bins}
    for k, v in got.items():
        assert k in ref, k
        assert float(v) == ref[k], (k, v, ref[k])
    # the KITTI settings file of the PL example (BASELINE config C3): ORB extractor, camera and the LSD feature count
    y = dict(re.findall(r"^([\w.]+)\s*:\s*([-+0-9.eE]+)\s*(?:#.*)?$", open("/root/reference/Examples/PL/PL_KITTI00-02.yaml", errors="replace").read(), re.M))
    want = {"ORBextractor.nFeatures": p.orb.nfeatures, "ORBextractor.scaleFactor": p.orb.scale_factor, "ORBextractor.nLevels": p.orb.nlevels,
            "ORBextractor.iniThFAST": p.orb.ini_th_fast, "ORBextractor.minThFAST": p.orb.min_th_fast, "Camera.fx": p.stereo.fx, "Camera.bf": p.stereo.bf,
            "lsd_nfeatures": p.line.lsd_nfeatures}
    for k, v in want.items():
        assert np.float32(y[k]) == np.float32(v), (k, v, y[k])


def test_is_in_frustum_matches_oracle():
    """Frame::isInFrustum over arrays (host arithmetic inside the product library, no device): every gate and every output bit against the
    oracle's restatement, on points scattered around the frustum so that each gate fires"""
    import orb_line_slam_amd as ola
    from orb_line_slam_amd import matcher
    import oracle_lib as oracle
    rng = np.random.default_rng(17)
    n = 20000
    kp = np.zeros(4, ola.KEYPOINT_DTYPE)
    f = ola.FrameView(kp, np.zeros((4, 32), np.uint8), None, np.float32(1.2) ** np.arange(8, dtype=np.float32), bounds=(0.0, 1241.0, 0.0, 376.0))
    f.mTcw = np.eye(4, dtype=np.float32)
    f.mTcw[:3, :3] = np.array([[0.9998, -0.01, 0.015], [0.0101, 0.9999, -0.005], [-0.0149, 0.0052, 0.9999]], np.float32)
    f.mTcw[:3, 3] = np.array([0.3, -0.1, 0.5], np.float32)
    world = np.stack([rng.uniform(-40, 40, n), rng.uniform(-12, 12, n), rng.uniform(-5, 60, n)], 1).astype(np.float32)
    d = np.linalg.norm(world, axis=1).astype(np.float32)
    normal = (world / np.maximum(d, 1e-3)[:, None] + rng.normal(0, 0.6, world.shape)).astype(np.float32)
    maxd = (d * rng.uniform(0.5, 4.0, n)).astype(np.float32)
    mind = (maxd / np.float32(1.2) ** 7 * rng.uniform(0.5, 1.5, n)).astype(np.float32)
    mp = ola.MapPointGeom(world, normal, maxd, mind, np.zeros((n, 32), np.uint8))
    for lim in (0.5, 0.0):
        oi, ol, oc, op = oracle.is_in_frustum(f, mp, lim)
        v = matcher.isInFrustum(f, mp, lim)
        assert np.array_equal(v.mbTrackInView, oi) and 0.02 * n < oi.sum() < 0.9 * n
        assert np.array_equal(v.mnTrackScaleLevel[oi], ol[oi]) and len(np.unique(ol[oi])) == 8
        assert np.array_equal(v.mTrackViewCos[oi].view(np.uint32), oc[oi].view(np.uint32))
        for k, arr in enumerate((v.mTrackProjX, v.mTrackProjY, v.mTrackProjXR)):
            assert np.array_equal(arr[oi].view(np.uint32), op[oi, k].view(np.uint32))


def test_sequence_loaders_follow_the_reference_layouts(tmp_path):
    """orb_line_slam_amd/sequence.py: the file lists of Examples/PL/PL_stereo_kitti.cc:130-160 and PL_stereo_euroc.cc:192-216, and the batch reader
    (left image of pair p at 2p, right at 2p + 1; grey files byte-for-byte).  Host code only."""
    from PIL import Image
    from orb_line_slam_amd import sequence
    rng = np.random.default_rng(3)
    w, h, n = 40, 24, 5
    imgs = rng.integers(0, 256, (2 * n, h, w), dtype=np.uint8)
    # KITTI: times.txt, image_0/%06d.png, image_1/%06d.png
    k = tmp_path / "kitti" / "00"
    (k / "image_0").mkdir(parents=True); (k / "image_1").mkdir()
    (k / "times.txt").write_text("".join("%e\n" % (0.1 * i) for i in range(n)) + "\n")
    for i in range(n):
        Image.fromarray(imgs[2 * i]).save(k / "image_0" / ("%06d.png" % i))
        Image.fromarray(imgs[2 * i + 1]).save(k / "image_1" / ("%06d.png" % i))
    l, r, t = sequence.load_images_kitti(str(k))
    assert len(l) == len(r) == len(t) == n and l[3].endswith("image_0/000003.png") and r[3].endswith("image_1/000003.png") and abs(t[3] - 0.3) < 1e-12
    seq = sequence.StereoSequence(str(k))
    assert (seq.width, seq.height, len(seq)) == (w, h, n)
    got = [b.copy() for b, _ in seq.batches(2)]
    assert [g.shape[0] for g in got] == [4, 4, 2] and np.array_equal(np.concatenate(got), imgs)
    # EuRoC: one stamp per line, <left>/<stamp>.png, <right>/<stamp>.png, seconds = stamp / 1e9; PGM files through the generic finder
    e = tmp_path / "mav0"
    (e / "cam0" / "data").mkdir(parents=True); (e / "cam1" / "data").mkdir(parents=True)
    stamps = [1403636579763555584 + 50000000 * i for i in range(n)]
    (tmp_path / "MH01.txt").write_text("\n".join(str(s) for s in stamps) + "\n")
    for i, s in enumerate(stamps):
        Image.fromarray(imgs[2 * i]).save(e / "cam0" / "data" / f"{s}.png")
        Image.fromarray(imgs[2 * i + 1]).save(e / "cam1" / "data" / f"{s}.png")
    l, r, t = sequence.load_images_euroc(str(e / "cam0" / "data"), str(e / "cam1" / "data"), str(tmp_path / "MH01.txt"))
    assert len(l) == n and l[1].endswith(f"cam0/data/{stamps[1]}.png") and abs(t[1] - stamps[1] / 1e9) < 1e-6
    seq = sequence.StereoSequence(left=l, right=r, times=t)
    assert np.array_equal(np.concatenate([b for b, _ in seq.batches(8)]), imgs)
    seq = sequence.StereoSequence(str(e), limit=3)
    assert len(seq) == 3 and np.array_equal(next(seq.batches(3))[0], imgs[:6])
    with pytest.raises(ValueError):
        sequence.StereoSequence(left=l, right=r[:-1])


def test_opencv_overloads_of_the_adaptor_type_check():
    """The ORBLINE_WITH_OPENCV blocks of include/orbline_adaptor.hpp (cv::InputArray / cv::OutputArray / cv::Mat / KeyLine overloads -- what the
    reference's Frame::ExtractORB / ExtractLine call, src/Frame.cc:350-364) had never been through a compiler: the image has no OpenCV.  They are
    type-checked here against tests/opencv_decl, a DECLARATION-ONLY stand-in written from the public OpenCV 3.4 API.  That stand-in is test
    infrastructure and pins nothing: the test proves the blocks parse and that the reference's call shapes resolve, not how OpenCV behaves."""
    out = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-Werror", "-DORBLINE_WITH_OPENCV", "-I" + os.path.join(ROOT, "tests", "opencv_decl"),
                          "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "adaptor_opencv_typecheck.cpp")],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert out.returncode == 0, out.stdout.decode()[-3000:]
