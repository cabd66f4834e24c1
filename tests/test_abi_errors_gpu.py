"""Error behaviour of the C ABI on a GPU box: bad arguments come back as status codes with a message, never as a crash or a silent no-op
(the reference throws std::runtime_error / asserts in the same situations: src/Frame.cc:146, src/ORBextractor.cc:1052)."""
import ctypes as C
import numpy as np
import pytest
from orb_line_slam_amd import _lib
from orb_line_slam_amd._lib import OLF_ERR_CAPACITY, OLF_ERR_INVALID, OLF_OK

pytestmark = pytest.mark.gpu


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_bad_arguments_are_reported():
    L = _lib.lib()
    p = _lib.default_params()
    h = C.c_void_p()
    assert L.olf_ctx_create(C.byref(p), 32, 32, 1, C.byref(h)) == OLF_ERR_INVALID and L.olf_last_error()        # below the minimum size
    assert L.olf_ctx_create(C.byref(p), 640, 480, 0, C.byref(h)) == OLF_ERR_INVALID
    assert L.olf_ctx_create(None, 640, 480, 1, C.byref(h)) == OLF_ERR_INVALID
    bad = _lib.default_params(); bad.orb.scale_factor = 1.0
    assert L.olf_ctx_create(C.byref(bad), 640, 480, 1, C.byref(h)) != OLF_OK                                 # the pyramid needs a factor > 1
    bad = _lib.default_params(); bad.line.lsd_scale = 0.3
    assert L.olf_ctx_create(C.byref(bad), 640, 480, 1, C.byref(h)) != OLF_OK                                 # LSD blur wider than 15 taps
    bad = _lib.default_params(); bad.orb.nfeatures = 13000                                                 # > 2040 key points on level 0: the octree spills to
    assert L.olf_ctx_create(C.byref(bad), 640, 480, 1, C.byref(h)) == OLF_OK                                  # global memory (tests/test_orb_gpu.py)
    L.olf_ctx_destroy(h)
    bad = _lib.default_params(); bad.orb.nfeatures = 250000
    assert L.olf_ctx_create(C.byref(bad), 640, 480, 1, C.byref(h)) == OLF_ERR_CAPACITY                       # > 32760 key points on level 0 (16-bit node ids)
    bad = _lib.default_params(); bad.orb.nfeatures = 2500; bad.orb.nlevels = 1
    assert L.olf_ctx_create(C.byref(bad), 640, 480, 1, C.byref(h)) == OLF_OK                                  # (round 1 refused this: 2500 key points on one level)
    L.olf_ctx_destroy(h)
    ctx = _lib.Context(p, 640, 480, 2)
    img = np.zeros((4, 480, 640), np.uint8)
    cap, lcap = ctx.orb_capacity, ctx.line_capacity
    kps, desc, cnt = np.zeros((4, cap), _lib.KEYPOINT_DTYPE), np.zeros((4, cap, 32), np.uint8), np.zeros(4, np.int32)
    assert L.olf_orb_extract(ctx.handle, _p(img), 3, _p(kps), _p(desc), _p(cnt)) == OLF_ERR_CAPACITY          # more images than max_images
    assert L.olf_orb_extract(ctx.handle, None, 1, _p(kps), _p(desc), _p(cnt)) == OLF_ERR_INVALID
    assert L.olf_orb_extract(ctx.handle, _p(img), 1, None, _p(desc), _p(cnt)) == OLF_ERR_INVALID
    assert L.olf_orb_extract(ctx.handle, _p(img), -1, _p(kps), _p(desc), _p(cnt)) != OLF_OK
    assert L.olf_orb_extract(ctx.handle, _p(img), 0, _p(kps), _p(desc), _p(cnt)) == OLF_OK                    # an empty batch is fine
    kls, ldesc = np.zeros((4, lcap), _lib.KEYLINE_DTYPE), np.zeros((4, lcap, 32), np.uint8)
    assert L.olf_line_extract(ctx.handle, _p(img), 5, _p(kls), _p(ldesc), _p(cnt)) == OLF_ERR_CAPACITY
    assert L.olf_line_extract(ctx.handle, _p(img), 1, None, _p(ldesc), _p(cnt)) == OLF_ERR_INVALID
    assert L.olf_stereo_points(ctx.handle, _p(img), 2, _p(kps), _p(desc), _p(cnt), None, None) != OLF_OK                  # null outputs
    d = np.zeros((3, 32), np.uint8); m = np.zeros(3, np.int32)
    assert L.olf_match_bf(ctx.handle, None, 3, _p(d), 3, C.c_float(0.9), 1, _p(m)) == OLF_ERR_INVALID
    assert L.olf_match_bf(ctx.handle, _p(d), -2, _p(d), 3, C.c_float(0.9), 1, _p(m)) == OLF_ERR_INVALID
    assert L.olf_match_bf(ctx.handle, _p(d), 0, _p(d), 3, C.c_float(0.9), 1, _p(m)) == OLF_OK
    assert L.olf_cvt_gray(ctx.handle, _p(img), 7, 1, _p(img)) == OLF_ERR_INVALID                               # unknown conversion code
    assert L.olf_distinctive_descriptors(ctx.handle, _p(d), _p(np.array([0, 3, 2], np.int32)), 2, _p(m)) == OLF_ERR_INVALID   # offsets decrease
    assert L.olf_voc_load_text(b"/nonexistent", C.byref(h)) == OLF_ERR_INVALID
    assert L.olf_voc_create(10, 0, 0, 0, 1, _p(m), _p(d), _p(d), _p(np.zeros(1)), C.byref(h)) == OLF_ERR_INVALID   # L must be >= 1
    # the per-frame searches: null views / outputs, an octave outside mvScaleFactors
    fv = _lib.FrameViewC()
    n1 = np.zeros(1, np.int32)
    assert L.olf_search_by_projection(ctx.handle, None, C.byref(fv), C.c_float(7), 0, 1, _p(m), _p(n1)) == OLF_ERR_INVALID
    assert L.olf_search_by_projection(ctx.handle, C.byref(fv), C.byref(fv), C.c_float(7), 0, 1, _p(m), _p(n1)) == OLF_ERR_INVALID   # no pose, no state
    assert L.olf_search_by_bow(ctx.handle, C.byref(fv), C.byref(fv), C.c_float(0.7), 1, None, _p(n1)) == OLF_ERR_INVALID
    assert L.olf_search_by_bow(ctx.handle, C.byref(fv), C.byref(fv), C.c_float(0.7), 1, _p(m), _p(n1)) == OLF_OK and n1[0] == 0      # two empty frames
    assert L.olf_search_local_map(ctx.handle, C.byref(fv), 3, None, None, None, None, None, None, None, C.c_float(1), C.c_float(0.8), _p(m),
                                  _p(n1)) == OLF_ERR_INVALID
    assert L.olf_last_error() is not None
    # the context is still usable after all of that
    k, dd, c = np.zeros((1, cap), _lib.KEYPOINT_DTYPE), np.zeros((1, cap, 32), np.uint8), np.zeros(1, np.int32)
    assert L.olf_orb_extract(ctx.handle, _p(np.full((1, 480, 640), 50, np.uint8)), 1, _p(k), _p(dd), _p(c)) == OLF_OK and c[0] == 0


def test_noise_beyond_the_old_segment_capacity_and_async_overflow_report(oracle):
    """Pure noise at lsd_scale 2 gives 9105 raw segments: rounds 1-3 refused it at run time (capacity 4096 / 8192, fuzz configuration #678).  The raw list is
    now sized by the working image (Ps / 48) and k_line_select sorts lists beyond its 64 KB of LDS in global memory: the top 100 must equal the oracle's.
    Then the asynchronous error report itself: a *_dev call that overflows a device buffer (here: a record buffer that is too small) does not fail by
    itself, the next olf_ctx_synchronize / olf_ctx_poll_status reports it -- once."""
    import torch
    L = _lib.lib()
    p = _lib.default_params()
    p.line.lsd_scale, p.line.min_line_length, p.line.lsd_quant, p.line.lsd_nfeatures = 2.0, 0.0, 1.0, 100
    w, h = 620, 470
    ctx = _lib.Context(p, w, h, 2)
    rng = np.random.default_rng(3)
    img_h = rng.integers(0, 256, (2, h, w), dtype=np.uint8)
    img = torch.from_numpy(img_h).cuda()
    lcap = ctx.line_capacity
    kls = torch.zeros(2 * lcap * _lib.KEYLINE_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    desc = torch.zeros(2 * lcap * 32, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(2, dtype=torch.int32, device="cuda")
    rc = L.olf_line_extract_dev(ctx.handle, C.c_void_p(img.data_ptr()), 2, C.c_void_p(kls.data_ptr()), C.c_void_p(desc.data_ptr()),
                                C.c_void_p(cnt.data_ptr()), None)
    assert rc == OLF_OK
    assert L.olf_ctx_synchronize(ctx.handle) == OLF_OK, L.olf_last_error()
    got = kls.cpu().numpy().view(_lib.KEYLINE_DTYPE).reshape(2, lcap)
    gd = desc.cpu().numpy().reshape(2, lcap, 32)
    for i in range(2):
        o = oracle.line_extract(img_h[i], p.line)
        assert int(cnt[i]) == len(o["kls"]) == 100
        assert np.array_equal(got[i, :100], o["kls"]) and np.array_equal(gd[i, :100], o["desc"])
    # an overflow is reported by the next synchronize, once
    cap = ctx.orb_capacity
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")
    t = {"kps": z((2, cap, 28), torch.uint8), "desc": z((2, cap, 32), torch.uint8), "counts": z((2,), torch.int32), "uright": z((1, cap), torch.float32),
         "depth": z((1, cap), torch.float32), "kls": z((2, lcap, 68), torch.uint8), "ldesc": z((2, lcap, 32), torch.uint8), "lcounts": z((2,), torch.int32),
         "lmatches12": z((1, lcap), torch.int32), "ldisp": z((1, lcap, 2), torch.float32), "lle": z((1, lcap, 3), torch.float64)}
    fb = _lib.FrameBuffers(*[t[k].data_ptr() for k in ("kps", "desc", "counts", "uright", "depth", "kls", "ldesc", "lcounts", "lmatches12", "ldisp", "lle")])
    s = torch.cuda.current_stream().cuda_stream
    assert L.olf_stereo_frames_dev(ctx.handle, C.c_void_p(img.data_ptr()), 1, C.byref(fb), C.c_void_p(s)) == OLF_OK
    small = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    nbytes = torch.zeros(1, dtype=torch.int64, device="cuda")
    assert L.olf_frames_pack_dev(ctx.handle, C.byref(fb), 1, C.c_void_p(small.data_ptr()), 2048, C.c_void_p(nbytes.data_ptr()), C.c_void_p(s)) == OLF_OK     # asynchronous: no error yet
    torch.cuda.synchronize()
    assert L.olf_ctx_synchronize(ctx.handle) == OLF_ERR_CAPACITY and b"overflow" in L.olf_last_error()
    assert L.olf_ctx_synchronize(ctx.handle) == OLF_OK                       # reported once, then cleared
    assert L.olf_ctx_poll_status(ctx.handle) == OLF_OK
