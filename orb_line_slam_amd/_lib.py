"""ctypes binding of liborbline_hip.so (include/orbline.h, include/orbline_types.h).

The library is built in-tree by `make -C orb_line_slam_amd/csrc` (or __graft_entry__.build()).
Importing this module never touches the GPU; creating a context does.  A missing library is a hard
error -- there is no pure-Python or CPU path behind these classes.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OLF_LIB_PATH", os.path.join(_HERE, "csrc", "liborbline_hip.so"))
SYNTH_PATH = os.path.join(_HERE, "csrc", "libolf_synth.so")

OLF_OK, OLF_ERR_INVALID, OLF_ERR_CAPACITY, OLF_ERR_HIP, OLF_ERR_NODEVICE = 0, -1, -2, -3, -4
DESC_BYTES = 32

# cv::KeyPoint (28 B) and cv::line_descriptor::KeyLine (68 B) record layouts
KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                           ("octave", "<i4"), ("class_id", "<i4")])
KEYLINE_DTYPE = np.dtype([("angle", "<f4"), ("class_id", "<i4"), ("octave", "<i4"), ("pt_x", "<f4"), ("pt_y", "<f4"),
                          ("response", "<f4"), ("size", "<f4"), ("startPointX", "<f4"), ("startPointY", "<f4"),
                          ("endPointX", "<f4"), ("endPointY", "<f4"), ("sPointInOctaveX", "<f4"),
                          ("sPointInOctaveY", "<f4"), ("ePointInOctaveX", "<f4"), ("ePointInOctaveY", "<f4"),
                          ("lineLength", "<f4"), ("numOfPixels", "<i4")])
assert KEYPOINT_DTYPE.itemsize == 28 and KEYLINE_DTYPE.itemsize == 68


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("conv_gauss_sum256", C.c_int32)]


class LineParams(C.Structure):
    _fields_ = [("lsd_nfeatures", C.c_int32), ("min_line_length", C.c_double), ("lsd_refine", C.c_int32),
                ("lsd_scale", C.c_double), ("lsd_sigma_scale", C.c_double), ("lsd_quant", C.c_double),
                ("lsd_ang_th", C.c_double), ("lsd_log_eps", C.c_double), ("lsd_density_th", C.c_double),
                ("lsd_n_bins", C.c_int32), ("conv_gauss_sum256", C.c_int32), ("conv_resize_exact", C.c_int32), ("conv_seed_order", C.c_int32),
                ("conv_libm_float", C.c_int32)]


class StereoParams(C.Structure):
    _fields_ = [("fx", C.c_float), ("bf", C.c_float), ("matching_s_ws", C.c_int32), ("line_sim_th", C.c_double),
                ("min_ratio_12_l", C.c_double), ("min_disp", C.c_double), ("line_horiz_th", C.c_double),
                ("stereo_overlap_th", C.c_double), ("ls_min_disp_ratio", C.c_double), ("best_lr_matches", C.c_int32),
                ("conv_eigen_recip", C.c_int32)]


class OlfParams(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("struct_size", C.c_uint32), ("orb", OrbParams), ("line", LineParams), ("stereo", StereoParams)]


class FrameBuffers(C.Structure):
    """olf_frame_buffers (include/orbline.h): device or host pointers of the fused stereo-frame entry"""
    _fields_ = [(n, C.c_void_p) for n in ("kps", "desc", "counts", "uright", "depth", "kls", "ldesc", "lcounts", "lmatches12",
                                          "ldisp", "lle")]


class FrameViewC(C.Structure):
    """olf_frame_view (include/orbline.h): the Frame / KeyFrame members read by the per-frame ORBmatcher searches"""
    _fields_ = ([(n, C.c_void_p) for n in ("keys", "desc", "uright")] + [("n", C.c_int32)] +
                [(n, C.c_void_p) for n in ("mp_valid", "mp_obs", "mp_bad", "mp_world", "mp_desc", "outlier", "Tcw")] +
                [(n, C.c_float) for n in ("fx", "fy", "cx", "cy", "mbf", "minX", "maxX", "minY", "maxY")] +
                [("scale_factors", C.c_void_p), ("n_levels", C.c_int32)] +
                [(n, C.c_void_p) for n in ("mp_maxd", "mp_mind", "fv_nodes", "fv_offsets", "fv_features")] + [("fv_n", C.c_int32)])


class OlfError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        super().__init__(f"{where}: status {code}: {last_error()}")


_lib = None


def _torch_runtime_first():
    """PyTorch ships its own copy of the HIP runtime.  A process that uses both this library and torch (bench.py, pipeline.py,
    distributed.py) must bring torch's runtime up first: the other order leaves this library without a visible device.  So when torch
    is installed it is imported and initialised before the library is loaded (set OLF_NO_TORCH=1 to skip: pure ctypes use)."""
    import importlib.util
    if os.environ.get("OLF_NO_TORCH") or importlib.util.find_spec("torch") is None:
        return
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


def lib():
    """The loaded C-ABI library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C orb_line_slam_amd/csrc` "
                              "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
        _torch_runtime_first()
        L = C.CDLL(LIB_PATH)
        L.olf_last_error.restype = C.c_char_p
        L.olf_ctx_create.argtypes = [C.POINTER(OlfParams), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.olf_ctx_destroy.argtypes = [C.c_void_p]
        L.olf_ctx_destroy.restype = None
        L.olf_ctx_synchronize.argtypes = [C.c_void_p]
        L.olf_ctx_poll_status.argtypes = [C.c_void_p]
        L.olf_orb_scale_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L.olf_orb_level_sizes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.olf_orb_capacity.argtypes = [C.c_void_p]
        L.olf_orb_extract_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.olf_orb_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.olf_orb_pyramid_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.olf_orb_debug_candidates.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.olf_stereo_points_dev.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6
        L.olf_stereo_points.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.olf_match_bf_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                       C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
        L.olf_match_bf.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.olf_knn2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.olf_hamming_matrix.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.olf_line_capacity.argtypes = [C.c_void_p]
        L.olf_line_extract_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4
        L.olf_line_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 3
        L.olf_lbd_compute.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 3
        L.olf_lsd_debug_scaled.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 3
        L.olf_stereo_lines_dev.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 7
        L.olf_stereo_lines.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6
        L.olf_stereo_frames_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(FrameBuffers), C.c_void_p]
        L.olf_stereo_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(FrameBuffers)]
        L.olf_match_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.olf_search_by_projection.argtypes = [C.c_void_p, C.POINTER(FrameViewC), C.POINTER(FrameViewC), C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.olf_search_by_projection_match12.argtypes = [C.c_void_p, C.POINTER(FrameViewC), C.POINTER(FrameViewC), C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.olf_search_by_bow.argtypes = [C.c_void_p, C.POINTER(FrameViewC), C.POINTER(FrameViewC), C.c_float, C.c_int, C.c_void_p, C.c_void_p]
        L.olf_search_for_initialization.argtypes = [C.c_void_p, C.POINTER(FrameViewC), C.POINTER(FrameViewC), C.c_void_p, C.c_int, C.c_float, C.c_int,
                                                    C.c_void_p, C.c_void_p]
        V = C.POINTER(FrameViewC)
        L.olf_is_in_frustum.argtypes = [V, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float] + [C.c_void_p] * 4
        L.olf_search_by_projection_kf.argtypes = [C.c_void_p, V, V, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.olf_search_by_bow_kf.argtypes = [C.c_void_p, V, V, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
        L.olf_search_for_triangulation.argtypes = [C.c_void_p, V, V, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.olf_fuse_search.argtypes = [C.c_void_p, V, C.c_int] + [C.c_void_p] * 6 + [C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.olf_fuse_search_sim3.argtypes = [C.c_void_p, V, C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_float, C.c_void_p, C.c_void_p]
        L.olf_search_by_projection_sim3.argtypes = [C.c_void_p, V, C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.olf_search_by_sim3.argtypes = [C.c_void_p, V, V, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.olf_search_local_map.argtypes = [C.c_void_p, C.POINTER(FrameViewC), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.olf_match_candidates_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.olf_cvt_gray.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.olf_remap_linear.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.olf_debug_lsd_waves.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.olf_debug_lsd_pool.argtypes = [C.c_void_p, C.c_int]
        L.olf_debug_lsd_groups.argtypes = [C.c_void_p, C.c_int]
        L.olf_debug_lsd_scatter.argtypes = [C.c_void_p, C.c_int]
        L.olf_debug_lsd_log_cap.argtypes = [C.c_void_p, C.c_int]
        L.olf_debug_seed_sort_mode.argtypes = [C.c_void_p, C.c_int]
        L.olf_debug_seed_sort.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int32)]
        L.olf_frames_pack_bound.argtypes = [C.c_void_p, C.c_int]
        L.olf_frames_pack_bound.restype = C.c_size_t
        L.olf_frames_pack_dev.argtypes = [C.c_void_p, C.POINTER(FrameBuffers), C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.olf_ctx_set_input_event.argtypes = [C.c_void_p, C.c_void_p]
        L.olf_ctx_set_deferred_join.argtypes = [C.c_void_p, C.c_int]
        L.olf_stereo_frames_join_dev.argtypes = [C.c_void_p, C.c_void_p]
        L.olf_stereo_points_mask_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.olf_debug_copy_bandwidth.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
        L.olf_debug_fdiv_sweep.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        L.olf_debug_sqrtq_sweep.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
        L.olf_debug_align_sweep.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        L.olf_debug_seed_sort_wide.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int32)]
        L.olf_distinctive_descriptors.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.olf_voc_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        L.olf_voc_load_text.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.olf_voc_destroy.argtypes = [C.c_void_p]
        L.olf_voc_destroy.restype = None
        L.olf_voc_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 6
        L.olf_bow_words_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.olf_bow_assemble.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.olf_search_by_bow_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.olf_bow_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.olf_profile_enable.argtypes = [C.c_void_p, C.c_int]
        L.olf_profile_reset.argtypes = [C.c_void_p]
        L.olf_profile_stage_name.restype = C.c_char_p
        L.olf_profile_stage_name.argtypes = [C.c_int]
        L.olf_profile_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def last_error():
    return lib().olf_last_error().decode()


def device_count():
    return int(lib().olf_device_count())


def default_params():
    p = OlfParams()
    rc = lib().olf_default_params(C.byref(p))
    if rc != OLF_OK:
        raise OlfError(rc, "olf_default_params")
    return p


def check(rc, where):
    if rc != OLF_OK:
        raise OlfError(rc, where)


def ptr(a):
    """void* of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One olf_ctx: fixed image size, parameters and per-call image capacity."""

    def __init__(self, params, width, height, max_images):
        self.params, self.width, self.height, self.max_images = params, int(width), int(height), int(max_images)
        h = C.c_void_p()
        check(lib().olf_ctx_create(C.byref(params), self.width, self.height, self.max_images, C.byref(h)), "olf_ctx_create")
        self.handle = h
        self.nlevels = params.orb.nlevels
        self.orb_capacity = int(lib().olf_orb_capacity(self.handle))
        self.line_capacity = int(lib().olf_line_capacity(self.handle))

    def close(self):
        if getattr(self, "handle", None):
            lib().olf_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def profile(self, on=True):
        check(lib().olf_profile_enable(self.handle, int(on)), "olf_profile_enable")
        check(lib().olf_profile_reset(self.handle), "olf_profile_reset")

    def profile_read(self):
        """{stage: (total_ms, calls)} accumulated since profile(True)"""
        n = lib().olf_profile_stage_count()
        ms = np.zeros(n, np.float64)
        calls = np.zeros(n, np.int32)
        check(lib().olf_profile_read(self.handle, ptr(ms), ptr(calls)), "olf_profile_read")
        return {lib().olf_profile_stage_name(i).decode(): (float(ms[i]), int(calls[i])) for i in range(n)}

    def synchronize(self):
        """wait for the context's streams; raises OlfError if a *_dev call issued since the last check overflowed a device buffer"""
        check(lib().olf_ctx_synchronize(self.handle), "olf_ctx_synchronize")

    def set_input_event(self, event):
        """olf_ctx_set_input_event: `event` a recorded torch.cuda.Event (or None): the line stream of olf_stereo_frames_dev then waits for it instead of
        forking from the caller's stream, so batch k + 1's LSD front runs beside batch k's tail.  The context keeps the raw handle only: keep the
        event alive while it is set."""
        check(lib().olf_ctx_set_input_event(self.handle, C.c_void_p(event.cuda_event) if event is not None else None), "olf_ctx_set_input_event")

    def poll_status(self):
        """the same check without waiting for any stream (the caller has synchronised its own)"""
        check(lib().olf_ctx_poll_status(self.handle), "olf_ctx_poll_status")
