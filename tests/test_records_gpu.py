"""GPU: the device packer of the trimmed frame record (csrc/records.hip, the multi-GPU payload) against its host mirror (records.py),
and the copy-kernel ceiling measurement."""
import ctypes as C
import numpy as np
import pytest
from orb_line_slam_amd import _lib, synth, records

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,npairs", [(640, 480, 3), (320, 240, 1)])
def test_device_pack_equals_host_pack(w, h, npairs):
    import torch
    dev = torch.device("cuda", 0)
    p = _lib.default_params()
    p.orb.nfeatures, p.line.lsd_nfeatures = 1000, 200
    ctx = _lib.Context(p, w, h, 2 * npairs)
    cap, lcap = ctx.orb_capacity, ctx.line_capacity
    imgs = torch.from_numpy(synth.stereo_batch(21, npairs, w, h)).to(dev)
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
    t = {"kps": z((2 * npairs, cap, 28), torch.uint8), "desc": z((2 * npairs, cap, 32), torch.uint8), "counts": z((2 * npairs,), torch.int32),
         "uright": z((npairs, cap), torch.float32), "depth": z((npairs, cap), torch.float32), "kls": z((2 * npairs, lcap, 68), torch.uint8),
         "ldesc": z((2 * npairs, lcap, 32), torch.uint8), "lcounts": z((2 * npairs,), torch.int32), "lmatches12": z((npairs, lcap), torch.int32),
         "ldisp": z((npairs, lcap, 2), torch.float32), "lle": z((npairs, lcap, 3), torch.float64)}
    fb = _lib.FrameBuffers(*[t[k].data_ptr() for k in ("kps", "desc", "counts", "uright", "depth", "kls", "ldesc", "lcounts", "lmatches12", "ldisp", "lle")])
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(L.olf_stereo_frames_dev(ctx.handle, imgs.data_ptr(), npairs, C.byref(fb), s), "olf_stereo_frames_dev")
    bound = L.olf_frames_pack_bound(ctx.handle, npairs)
    dst = torch.full((bound,), 0xEE, dtype=torch.uint8, device=dev)
    nbytes = torch.zeros(1, dtype=torch.int64, device=dev)
    _lib.check(L.olf_frames_pack_dev(ctx.handle, C.byref(fb), npairs, dst.data_ptr(), bound, nbytes.data_ptr(), s), "olf_frames_pack_dev")
    torch.cuda.synchronize()
    ctx.synchronize()
    host = {k: v.cpu().numpy() for k, v in t.items()}
    want = records.pack_records(host, host["counts"], host["lcounts"])
    n = int(nbytes.item())
    assert n == len(want) <= bound
    got = dst[:n].cpu().numpy().tobytes()
    # padding bytes between sections are not written by the device packer: compare through the parser, then the payload bytes
    pg, pw = records.parse_records(got), records.parse_records(want)
    assert pg["n_pairs"] == npairs and pg["counts"].sum() > 0 and pg["lcounts"].sum() > 0
    for k in ("counts", "lcounts") + tuple(s_[0] for s_ in records.SECTIONS):
        assert np.array_equal(pg[k], pw[k]), k
    assert records.merge_records([got]) == want
    # a destination that is too small is reported, not overrun
    small = torch.full((4096,), 0xEE, dtype=torch.uint8, device=dev)
    _lib.check(L.olf_frames_pack_dev(ctx.handle, C.byref(fb), npairs, small.data_ptr(), 2048, nbytes.data_ptr(), s), "olf_frames_pack_dev")
    torch.cuda.synchronize()
    with pytest.raises(_lib.OlfError):
        ctx.synchronize()
    assert (small[2048:] == 0xEE).all()
    ctx.close()


def test_copy_bandwidth_is_measured():
    p = _lib.default_params()
    ctx = _lib.Context(p, 640, 480, 2)
    g = C.c_double()
    _lib.check(_lib.lib().olf_debug_copy_bandwidth(ctx.handle, 1 << 30, 10, C.byref(g)), "olf_debug_copy_bandwidth")
    assert 500.0 < g.value < 9000.0, g.value
    ctx.close()
