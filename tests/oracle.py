"""ctypes binding of oracle/liborb_oracle.so — the CPU restatement used ONLY as the checker."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KP = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
               ("octave", "<i4"), ("class_id", "<i4")])
BLUR_CV_4_0_0, BLUR_CV_4_2_PLUS = 0, 1
_LIB = None
_P = C.c_void_p


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "liborb_oracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        L = C.CDLL(path)
        L.orc_cv_round_f.restype = C.c_int
        L.orc_cv_round_f.argtypes = [C.c_float]
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_resize_linear_u8.restype = None
        L.orc_resize_linear_u8.argtypes = [_P, C.c_int, C.c_int, C.c_long, _P, C.c_int, C.c_int, C.c_long]
        L.orc_gaussian7_u8.restype = None
        L.orc_gaussian7_u8.argtypes = [_P, C.c_int, C.c_int, C.c_long, _P, C.c_long, C.c_int]
        L.orc_fast9_16.restype = C.c_int
        L.orc_fast9_16.argtypes = [_P, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int, _P, C.c_int]
        L.orc_distribute_octree.restype = C.c_int
        L.orc_distribute_octree.argtypes = [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int]
        L.orc_create.restype = _P
        L.orc_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_destroy.restype = None
        L.orc_destroy.argtypes = [_P]
        L.orc_set_trig_mode.restype = None
        L.orc_set_trig_mode.argtypes = [C.c_int]
        for name in ("orc_extract", "orc_extract_fast"):
            fn = getattr(L, name)
            fn.restype = C.c_int
            fn.argtypes = [_P, _P, C.c_int, C.c_int, C.c_long, _P, _P, C.c_int]
        L.orc_levels.restype = C.c_int
        L.orc_levels.argtypes = [_P]
        L.orc_scale_factors.restype = None
        L.orc_scale_factors.argtypes = [_P, _P, _P, _P, _P]
        L.orc_features_per_level.restype = None
        L.orc_features_per_level.argtypes = [_P, _P]
        L.orc_umax.restype = None
        L.orc_umax.argtypes = [_P, _P]
        L.orc_level_size.restype = C.c_int
        L.orc_level_size.argtypes = [_P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        for name in ("orc_level_padded", "orc_level_image", "orc_level_blurred"):
            fn = getattr(L, name)
            fn.restype = C.c_int
            fn.argtypes = [_P, C.c_int, _P, C.c_long]
        for name in ("orc_level_candidates", "orc_level_selected"):
            fn = getattr(L, name)
            fn.restype = C.c_int
            fn.argtypes = [_P, C.c_int, _P, C.c_int]
        for name in ("orc_hamming256", "orc_hamming256_popcnt"):
            fn = getattr(L, name)
            fn.restype = C.c_int
            fn.argtypes = [_P, _P]
        L.orc_knn2.restype = None
        L.orc_knn2.argtypes = [_P, C.c_long, _P, C.c_long, C.c_int, C.c_int, _P, _P, _P]
        L.orc_ratio_accept.restype = C.c_int
        L.orc_ratio_accept.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]
        L.orc_stereo_band_match.restype = C.c_int
        L.orc_stereo_band_match.argtypes = [_P, _P, C.c_int, _P, _P, C.c_int, C.c_int, _P, C.c_int,
                                            C.c_float, C.c_float, _P, _P]
        L.orc_compute_stereo_matches.restype = C.c_int
        L.orc_compute_stereo_matches.argtypes = [_P, _P, _P, _P, C.c_int, _P, _P, C.c_int, C.c_float, C.c_float, _P, _P]
        L.orc_undistort_points.restype = None
        L.orc_undistort_points.argtypes = [_P, C.c_int, _P, _P]
        L.orc_assign_grid.restype = None
        L.orc_assign_grid.argtypes = [_P, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, _P, _P]
        L.orc_search_for_initialization.restype = C.c_int
        L.orc_search_for_initialization.argtypes = [_P, _P, C.c_int, _P, _P, C.c_int, C.c_float, C.c_float, C.c_float,
                                                    C.c_float, _P, C.c_int, C.c_float, C.c_int, _P]
        L.orc_search_by_bow.restype = C.c_int
        L.orc_search_by_bow.argtypes = [C.c_int, _P, _P, _P, C.c_int, _P, _P, _P, C.c_int, _P, _P, _P, C.c_int,
                                        _P, _P, _P, C.c_int, C.c_float, C.c_int, _P]
        L.orc_search_by_projection_frame.restype = C.c_int
        L.orc_search_by_projection_frame.argtypes = [_P, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P, C.c_int,
                                                     C.c_float, C.c_float, C.c_float, C.c_float, _P, C.c_int,
                                                     C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, _P]
        L.orc_search_by_projection_points.restype = C.c_int
        L.orc_search_by_projection_points.argtypes = [_P, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P, C.c_int,
                                                      C.c_float, C.c_float, C.c_float, C.c_float, _P, C.c_int,
                                                      C.c_float, C.c_float, _P]
        _LIB = L
    return _LIB


def ptr(a):
    return a.ctypes.data_as(_P)


def set_trig_mode(libm_float):
    """False (default): descriptor steering uses the correctly rounded cos / sin; True: this runtime's cosf / sinf."""
    lib().orc_set_trig_mode(1 if libm_float else 0)


def resize(src, dw, dh):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.empty((dh, dw), dtype=np.uint8)
    lib().orc_resize_linear_u8(ptr(src), src.shape[1], src.shape[0], src.strides[0], ptr(dst), dw, dh, dst.strides[0])
    return dst


def gaussian7(src, variant=BLUR_CV_4_0_0):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.empty_like(src)
    lib().orc_gaussian7_u8(ptr(src), src.shape[1], src.shape[0], src.strides[0], ptr(dst), dst.strides[0], variant)
    return dst


def fast(img, threshold, nms=True, cap=200000):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty(cap, dtype=KP)
    n = lib().orc_fast9_16(ptr(img), img.shape[1], img.shape[0], img.strides[0], threshold, 1 if nms else 0, ptr(out), cap)
    assert n <= cap
    return out[:n].copy()


def distribute_octree(kps, minX, maxX, minY, maxY, N):
    kps = np.ascontiguousarray(kps, dtype=KP)
    out = np.empty(max(len(kps), 1), dtype=KP)
    n = lib().orc_distribute_octree(ptr(kps), len(kps), minX, maxX, minY, maxY, N, ptr(out), len(out))
    return out[:n].copy()


class Extractor:
    def __init__(self, nfeatures=1000, scale=1.2, nlevels=8, ini=20, mn=7, blur=BLUR_CV_4_0_0):
        self.h = lib().orc_create(nfeatures, scale, nlevels, ini, mn, blur)
        assert self.h
        self.nlevels = nlevels
        self.cap = nfeatures + 300 * nlevels + 64   # a level can keep 4 * round(width / height) keypoints whatever its quota

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_destroy(self.h)
            self.h = None

    def extract(self, img, fast=False):
        kps = np.empty(self.cap, dtype=KP)
        desc = np.zeros((self.cap, 32), dtype=np.uint8)
        if img is None or img.size == 0:
            n = lib().orc_extract(self.h, None, 0, 0, 0, ptr(kps), ptr(desc), self.cap)
            return kps[:n], desc[:n]
        img = np.ascontiguousarray(img, dtype=np.uint8)
        fn = lib().orc_extract_fast if fast else lib().orc_extract
        n = fn(self.h, ptr(img), img.shape[1], img.shape[0], img.strides[0], ptr(kps), ptr(desc), self.cap)
        assert 0 <= n <= self.cap
        return kps[:n].copy(), desc[:n].copy()

    def scale_factors(self):
        out = [np.empty(self.nlevels, dtype=np.float32) for _ in range(4)]
        lib().orc_scale_factors(self.h, *[ptr(o) for o in out])
        return out

    def features_per_level(self):
        out = np.empty(self.nlevels, dtype=np.int32)
        lib().orc_features_per_level(self.h, ptr(out))
        return out

    def umax(self):
        out = np.empty(16, dtype=np.int32)
        lib().orc_umax(self.h, ptr(out))
        return out

    def level_size(self, level):
        w, h = C.c_int(0), C.c_int(0)
        assert lib().orc_level_size(self.h, level, C.byref(w), C.byref(h)) == 0
        return w.value, h.value

    def level_image(self, level, padded=False):
        w, h = self.level_size(level)
        pad = 38 if padded else 0
        out = np.empty((h + pad, w + pad), dtype=np.uint8)
        fn = lib().orc_level_padded if padded else lib().orc_level_image
        assert fn(self.h, level, ptr(out), out.strides[0]) == 0
        return out

    def level_blurred(self, level):
        w, h = self.level_size(level)
        out = np.empty((h, w), dtype=np.uint8)
        rc = lib().orc_level_blurred(self.h, level, ptr(out), out.strides[0])
        return out if rc == 0 else None

    def _list(self, fn, level):
        out = np.empty(400000, dtype=KP)
        n = fn(self.h, level, ptr(out), len(out))
        assert 0 <= n <= len(out)
        return out[:n].copy()

    def level_candidates(self, level):
        return self._list(lib().orc_level_candidates, level)

    def level_selected(self, level):
        return self._list(lib().orc_level_selected, level)


def hamming(a, b, popcnt=False):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    return (lib().orc_hamming256_popcnt if popcnt else lib().orc_hamming256)(ptr(a), ptr(b))


def knn2(q, db, init=256, popcnt=True):
    q = np.ascontiguousarray(q, dtype=np.uint8).reshape(-1, 32)
    db = np.ascontiguousarray(db, dtype=np.uint8).reshape(-1, 32)
    idx = np.empty(len(q), dtype=np.int32)
    d1 = np.empty(len(q), dtype=np.int32)
    d2 = np.empty(len(q), dtype=np.int32)
    lib().orc_knn2(ptr(q), len(q), ptr(db), len(db), init, 1 if popcnt else 0, ptr(idx), ptr(d1), ptr(d2))
    return idx, d1, d2


def stereo_band_match(kl, dl, kr, dr, rows, scale, bf, b):
    kl = np.ascontiguousarray(kl, dtype=KP)
    kr = np.ascontiguousarray(kr, dtype=KP)
    dl = np.ascontiguousarray(dl, dtype=np.uint8)
    dr = np.ascontiguousarray(dr, dtype=np.uint8)
    scale = np.ascontiguousarray(scale, dtype=np.float32)
    bi = np.empty(len(kl), dtype=np.int32)
    bd = np.empty(len(kl), dtype=np.int32)
    rc = lib().orc_stereo_band_match(ptr(kl), ptr(dl), len(kl), ptr(kr), ptr(dr), len(kr), rows, ptr(scale),
                                     len(scale), bf, b, ptr(bi), ptr(bd))
    assert rc == 0
    return bi, bd


def compute_stereo_matches(ex_left, ex_right, kl, dl, kr, dr, bf, b):
    """ex_left / ex_right: oracle Extractor objects that just extracted the two images."""
    kl = np.ascontiguousarray(kl, dtype=KP)
    kr = np.ascontiguousarray(kr, dtype=KP)
    dl = np.ascontiguousarray(dl, dtype=np.uint8)
    dr = np.ascontiguousarray(dr, dtype=np.uint8)
    ur = np.empty(len(kl), dtype=np.float32)
    z = np.empty(len(kl), dtype=np.float32)
    kept = lib().orc_compute_stereo_matches(ex_left.h, ex_right.h, ptr(kl), ptr(dl), len(kl), ptr(kr), ptr(dr), len(kr),
                                            bf, b, ptr(ur), ptr(z))
    assert kept >= 0
    return ur, z, kept


def undistort_points(xy, K4, D5):
    xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2).copy()
    K4 = np.ascontiguousarray(K4, dtype=np.float32)
    D5 = np.ascontiguousarray(D5, dtype=np.float32)
    lib().orc_undistort_points(ptr(xy), len(xy), ptr(K4), ptr(D5))
    return xy


def assign_grid(kps, bounds):
    kps = np.ascontiguousarray(kps, dtype=KP)
    start = np.empty(64 * 48 + 1, dtype=np.int32)
    items = np.full(max(len(kps), 1), -1, dtype=np.int32)
    lib().orc_assign_grid(ptr(kps), len(kps), bounds[0], bounds[1], bounds[2], bounds[3], ptr(start), ptr(items))
    return start, items[:start[-1]]


def search_for_initialization(k1, d1, k2, d2, bounds, prev, window=100, nnratio=0.9, check_ori=True):
    k1 = np.ascontiguousarray(k1, dtype=KP)
    k2 = np.ascontiguousarray(k2, dtype=KP)
    d1 = np.ascontiguousarray(d1, dtype=np.uint8)
    d2 = np.ascontiguousarray(d2, dtype=np.uint8)
    prev = np.ascontiguousarray(prev, dtype=np.float32).copy()
    out = np.empty(len(k1), dtype=np.int32)
    n = lib().orc_search_for_initialization(ptr(k1), ptr(d1), len(k1), ptr(k2), ptr(d2), len(k2), bounds[0], bounds[1],
                                            bounds[2], bounds[3], ptr(prev), window, nnratio, 1 if check_ori else 0,
                                            ptr(out))
    return n, out, prev


def search_by_bow(k1, d1, valid1, fv1, k2, d2, fv2, valid2=None, nnratio=0.7, check_ori=True):
    """fvK = (nodes uint32 ascending, start int32, items uint32)."""
    k1 = np.ascontiguousarray(k1, dtype=KP)
    k2 = np.ascontiguousarray(k2, dtype=KP)
    d1 = np.ascontiguousarray(d1, dtype=np.uint8)
    d2 = np.ascontiguousarray(d2, dtype=np.uint8)
    v1 = np.ascontiguousarray(valid1, dtype=np.uint8)
    v2 = None if valid2 is None else np.ascontiguousarray(valid2, dtype=np.uint8)
    a = [np.ascontiguousarray(fv1[0], np.uint32), np.ascontiguousarray(fv1[1], np.int32), np.ascontiguousarray(fv1[2], np.uint32)]
    b = [np.ascontiguousarray(fv2[0], np.uint32), np.ascontiguousarray(fv2[1], np.int32), np.ascontiguousarray(fv2[2], np.uint32)]
    mode = 0 if v2 is None else 1
    out = np.empty(len(k2) if mode == 0 else len(k1), dtype=np.int32)
    n = lib().orc_search_by_bow(mode, ptr(k1), ptr(d1), ptr(v1), len(k1), ptr(a[0]), ptr(a[1]), ptr(a[2]), len(a[0]),
                                ptr(k2), ptr(d2), None if v2 is None else ptr(v2), len(k2), ptr(b[0]), ptr(b[1]),
                                ptr(b[2]), len(b[0]), nnratio, 1 if check_ori else 0, ptr(out))
    return n, out


def search_by_projection_frame(uv, inv_z, last_octave, last_angle, desc_mp, has_obs, k2, d2, bounds, scale_factors,
                               th, mbf=0.0, direction=0, u_right=None, occupied=None, check_ori=True, max_dist=100):
    uv = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
    iz = np.ascontiguousarray(inv_z, dtype=np.float32)
    lo = np.ascontiguousarray(last_octave, dtype=np.int32)
    la = np.ascontiguousarray(last_angle, dtype=np.float32)
    dm = np.ascontiguousarray(desc_mp, dtype=np.uint8).reshape(-1, 32)
    ho = np.ascontiguousarray(has_obs, dtype=np.uint8)
    k2 = np.ascontiguousarray(k2, dtype=KP)
    d2 = np.ascontiguousarray(d2, dtype=np.uint8)
    sf = np.ascontiguousarray(scale_factors, dtype=np.float32)
    ur = None if u_right is None else np.ascontiguousarray(u_right, dtype=np.float32)
    oc = None if occupied is None else np.ascontiguousarray(occupied, dtype=np.uint8)
    out = np.empty(len(k2), dtype=np.int32)
    n = lib().orc_search_by_projection_frame(ptr(uv), ptr(iz), ptr(lo), ptr(la), ptr(dm), ptr(ho), len(uv), ptr(k2),
                                             ptr(d2), None if ur is None else ptr(ur),
                                             None if oc is None else ptr(oc), len(k2), bounds[0], bounds[1],
                                             bounds[2], bounds[3], ptr(sf), len(sf), th, mbf, direction, max_dist,
                                             1 if check_ori else 0, ptr(out))
    return n, out


def search_by_projection_points(proj_xy, view_cos, level, desc_mp, has_obs, k2, d2, bounds, scale_factors, th=1.0,
                                nnratio=0.8, proj_xr=None, u_right=None, occupied=None):
    xy = np.ascontiguousarray(proj_xy, dtype=np.float32).reshape(-1, 2)
    vc = np.ascontiguousarray(view_cos, dtype=np.float32)
    lv = np.ascontiguousarray(level, dtype=np.int32)
    dm = np.ascontiguousarray(desc_mp, dtype=np.uint8).reshape(-1, 32)
    ho = np.ascontiguousarray(has_obs, dtype=np.uint8)
    k2 = np.ascontiguousarray(k2, dtype=KP)
    d2 = np.ascontiguousarray(d2, dtype=np.uint8)
    sf = np.ascontiguousarray(scale_factors, dtype=np.float32)
    xr = np.zeros(len(xy), np.float32) if proj_xr is None else np.ascontiguousarray(proj_xr, dtype=np.float32)
    ur = None if u_right is None else np.ascontiguousarray(u_right, dtype=np.float32)
    oc = None if occupied is None else np.ascontiguousarray(occupied, dtype=np.uint8)
    out = np.empty(len(k2), dtype=np.int32)
    n = lib().orc_search_by_projection_points(ptr(xy), ptr(xr), ptr(vc), ptr(lv), ptr(dm), ptr(ho), len(xy), ptr(k2),
                                              ptr(d2), None if ur is None else ptr(ur),
                                              None if oc is None else ptr(oc), len(k2), bounds[0], bounds[1],
                                              bounds[2], bounds[3], ptr(sf), len(sf), th, nnratio, ptr(out))
    return n, out

