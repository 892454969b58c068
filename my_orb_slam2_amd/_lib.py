"""ctypes binding of liborbx.so (include/orbx.h).  No fallback: a missing library is an error."""
import ctypes as C
import os

import numpy as np

BLUR_CV_4_0_0 = 0
BLUR_CV_4_2_PLUS = 1

# cv::KeyPoint layout (28 bytes)
KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KEYPOINT_DTYPE.itemsize == 28


class OrbxError(RuntimeError):
    def __init__(self, code, where):
        msg = lib().orbx_last_error().decode(errors="replace")
        super().__init__("%s failed with code %d: %s" % (where, code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("device", C.c_int32),
                ("max_width", C.c_int32), ("max_height", C.c_int32), ("max_batch", C.c_int32),
                ("blur_taps", C.c_int32)]


class Camera(C.Structure):
    """orbx_camera: Camera.fx fy cx cy k1 k2 p1 p2 k3 of the reference's settings file."""
    _fields_ = [(n, C.c_float) for n in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3")]


def lib_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "liborbx.so")


_LIB = None

# name -> (restype, argtypes); every symbol include/orbx.h declares
_P = C.c_void_p
SIGNATURES = {
    "orbx_last_error": (C.c_char_p, []),
    "orbx_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "orbx_destroy": (None, [_P]),
    "orbx_get_levels": (C.c_int, [_P]),
    "orbx_get_scale_factor": (C.c_float, [_P]),
    "orbx_get_scale_factors": (C.c_int, [_P, _P]),
    "orbx_get_inverse_scale_factors": (C.c_int, [_P, _P]),
    "orbx_get_scale_sigma_squares": (C.c_int, [_P, _P]),
    "orbx_get_inverse_scale_sigma_squares": (C.c_int, [_P, _P]),
    "orbx_get_features_per_level": (C.c_int, [_P, _P]),
    "orbx_get_umax": (C.c_int, [_P, _P]),
    "orbx_keypoint_capacity": (C.c_int, [_P]),
    "orbx_extract": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_size_t, _P, _P, C.c_int, C.POINTER(C.c_int)]),
    "orbx_extract_batch_device": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t,
                                            _P, _P, C.c_int, _P, _P]),
    "orbx_extract_batch": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t,
                                     _P, _P, C.c_int, _P]),
    "orbx_synchronize": (C.c_int, [_P]),
    "orbx_set_profiling": (C.c_int, [_P, C.c_int]),
    "orbx_get_stage_ms": (C.c_int, [_P, _P]),
    "orbx_level_size": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "orbx_get_level": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_size_t]),
    "orbx_debug_check_pyramid_plan": (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.POINTER(C.c_int)]),
    "orbx_debug_keypoint_blur": (C.c_int, [_P, C.c_int, _P, C.c_int, C.POINTER(C.c_int)]),
    "orbx_debug_candidates": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.POINTER(C.c_int)]),
    "orbx_debug_selected": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.POINTER(C.c_int)]),
    "orbx_hamming256": (C.c_int, [_P, _P]),
    "orbx_matcher_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "orbx_matcher_destroy": (None, [_P]),
    "orbx_hamming_knn2_device": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, C.c_int32, C.c_int64,
                                           _P, _P, _P, _P]),
    "orbx_hamming_knn2_noalloc_device": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, C.c_int32, C.c_int64,
                                                   _P, _P, _P, _P]),
    "orbx_hamming_knn2": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, C.c_int32, _P, _P, _P]),
    "orbx_knn2_merge_device": (C.c_int, [_P, C.c_int, C.c_int64, _P, _P, _P, C.c_int32, _P, _P, _P, _P]),
    "orbx_hamming_knn2_packed_device": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, C.c_int32, C.c_int64, _P, _P]),
    "orbx_knn2_merge_packed_device": (C.c_int, [_P, C.c_int, C.c_int64, _P, C.c_int32, _P, _P, _P, _P]),
    "orbx_ratio_test_device": (C.c_int, [_P, C.c_int64, _P, _P, C.c_int32, C.c_int, C.c_float, _P, _P]),
    "orbx_stereo_band_match_device": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P,
                                                C.c_int, C.c_float, C.c_float, _P, _P, _P]),
    "orbx_compute_stereo_matches_device": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P,
                                                     C.c_float, C.c_float, _P, _P, _P]),
    "orbx_matcher_synchronize": (C.c_int, [_P]),
    "orbx_matcher_get_stream": (C.c_int, [_P, C.POINTER(_P)]),
    "orbx_get_stream": (C.c_int, [_P, C.POINTER(_P)]),
    "orbx_search_for_initialization": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, C.c_int, _P, _P, C.c_float, C.c_float,
                                                 C.c_float, C.c_float, _P, C.c_int, C.c_float, C.c_int, _P,
                                                 C.POINTER(C.c_int)]),
    "orbx_search_for_initialization_device": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, _P, C.c_int, _P, _P, C.c_float,
                                                        C.c_float, C.c_float, C.c_float, _P, C.c_int, C.c_float,
                                                        C.c_int, _P, C.POINTER(C.c_int)]),
    "orbx_search_by_bow_device": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int, _P, _P, _P, C.c_int, _P, _P, _P, C.c_int,
                                            _P, _P, _P, C.c_int, C.c_float, C.c_int, _P, C.POINTER(C.c_int)]),
    "orbx_search_by_projection_frame": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P, C.c_int, _P, _P,
                                                  C.c_float, C.c_float, C.c_float, C.c_float, _P, C.c_int, C.c_float,
                                                  C.c_float, C.c_int, C.c_int, C.c_int, _P, C.POINTER(C.c_int)]),
    "orbx_search_by_projection_points": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P, C.c_int, _P, _P,
                                                   C.c_float, C.c_float, C.c_float, C.c_float, _P, C.c_int, C.c_float,
                                                   C.c_float, _P, C.POINTER(C.c_int)]),
    "orbx_search_by_bow": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int, _P, _P, _P, C.c_int, _P, _P, _P, C.c_int,
                                     _P, _P, _P, C.c_int, C.c_float, C.c_int, _P, C.POINTER(C.c_int)]),
    "orbx_image_bounds": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P]),
    "orbx_undistort_assign_grid_device": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, C.c_float, C.c_float,
                                                    C.c_float, C.c_float, _P, _P, _P, _P]),
    "orbx_synth_frame": (C.c_int, [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   _P, C.c_size_t]),
}


def lib():
    """Loads liborbx.so once. Raises if it has not been built (see __graft_entry__.build)."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise ImportError("liborbx.so not built: run `make -C my_orb_slam2_amd/csrc` "
                              "(expected at %s)" % path)
        # One HIP runtime per process: PyTorch ships its own libamdhip64.so.7 and liborbx.so is linked
        # against the same SONAME. If torch is loaded first, liborbx binds to torch's copy and the two
        # share devices, streams and allocations; loaded the other way round, the process ends up with
        # two runtimes and torch then reports "No HIP GPUs are available". So load torch first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        handle = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = handle
    return _LIB


def check(code, where):
    if code != 0:
        raise OrbxError(code, where)


def np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)
