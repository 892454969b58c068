"""Host mirror of the Hamming half of ORB_SLAM2::ORBmatcher (my_orb_slam2/include/ORBmatcher.h)
and of the descriptor band search of Frame::ComputeStereoMatches (src/Frame.cc:503-586)."""
import collections
import ctypes as C

import numpy as np

from ._lib import Camera, check, lib, np_ptr

# include/PresetParameters.h:11-13
TH_HIGH = 100
TH_LOW = 50
HISTO_LENGTH = 30


class ORBmatcher:
    """Static distance + default ratio of the reference's matcher (include/ORBmatcher.h:44-47)."""

    def __init__(self, nnratio=0.6, checkOri=True):
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)

    @staticmethod
    def DescriptorDistance(a, b):
        """ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1698-1714) on two 32-byte rows."""
        a = np.ascontiguousarray(a, dtype=np.uint8).reshape(32)
        b = np.ascontiguousarray(b, dtype=np.uint8).reshape(32)
        return lib().orbx_hamming256(np_ptr(a), np_ptr(b))


class HammingMatcher:
    """Brute-force best / second-best search on the GPU (update rule of src/ORBmatcher.cc:204-229)."""

    def __init__(self, device=0):
        self._m = C.c_void_p()
        check(lib().orbx_matcher_create(int(device), C.byref(self._m)), "orbx_matcher_create")
        self._device = int(device)
        self._ext_stream = None
        self._inflight = collections.deque()
        self.device = int(device)

    def close(self):
        if getattr(self, "_m", None) and self._m.value:
            if self._inflight:
                lib().orbx_matcher_synchronize(self._m)
                self._inflight.clear()
            self._ext_stream = None
            lib().orbx_matcher_destroy(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def knn2(self, query, db, init=256):
        """Host arrays (nq, 32) and (ndb, 32) uint8 -> idx int64, d1, d2 int32."""
        query = np.ascontiguousarray(query, dtype=np.uint8).reshape(-1, 32)
        db = np.ascontiguousarray(db, dtype=np.uint8).reshape(-1, 32)
        nq = query.shape[0]
        idx = np.empty(nq, dtype=np.int64)
        d1 = np.empty(nq, dtype=np.int32)
        d2 = np.empty(nq, dtype=np.int32)
        check(lib().orbx_hamming_knn2(self._m, np_ptr(query), nq, np_ptr(db), db.shape[0], int(init),
                                      np_ptr(idx), np_ptr(d1), np_ptr(d2)), "orbx_hamming_knn2")
        return idx, d1, d2

    def knn2_device(self, d_query, d_db, d_idx, d_d1, d_d2, init=256, index_base=0, stream=None, noalloc=False):
        """torch CUDA tensors: (nq, 32) / (ndb, 32) uint8 in, int64 / int32 / int32 (nq,) out.
        noalloc: the entry point that keeps its partial results in the library's fixed device array."""
        sp = C.c_void_p(stream) if stream else None
        self._before(stream)
        fn = lib().orbx_hamming_knn2_noalloc_device if noalloc else lib().orbx_hamming_knn2_device
        check(fn(self._m, C.c_void_p(d_query.data_ptr()), d_query.shape[0],
                                             C.c_void_p(d_db.data_ptr()), d_db.shape[0], int(init),
                                             int(index_base), C.c_void_p(d_idx.data_ptr()),
                                             C.c_void_p(d_d1.data_ptr()), C.c_void_p(d_d2.data_ptr()), sp),
              "orbx_hamming_knn2_noalloc_device" if noalloc else "orbx_hamming_knn2_device")
        self._hold(stream, d_query, d_db, d_idx, d_d1, d_d2)

    def knn2_packed_device(self, d_query, d_db, d_hits, init=256, index_base=0, stream=None):
        """The scan with one packed 16-byte record {int32 d1, d2; int64 idx} per query: d_hits is an (nq, 2) int64
        (or (nq, 16) uint8) CUDA tensor, the unit one all-gather moves when the database is sharded over GPUs."""
        assert d_hits.is_contiguous() and d_hits.numel() * d_hits.element_size() == 16 * d_query.shape[0]
        sp = C.c_void_p(stream) if stream else None
        self._before(stream)
        check(lib().orbx_hamming_knn2_packed_device(self._m, C.c_void_p(d_query.data_ptr()), d_query.shape[0],
                                                    C.c_void_p(d_db.data_ptr()), d_db.shape[0], int(init),
                                                    int(index_base), C.c_void_p(d_hits.data_ptr()), sp),
              "orbx_hamming_knn2_packed_device")
        self._hold(stream, d_query, d_db, d_hits)

    def merge_packed_device(self, d_hits, nparts, o_idx, o_d1, o_d2, init=256, stream=None):
        """Fold the gathered records of `nparts` consecutive database shards (part p of query i at record
        p * nq + i) into the result of one scan over the whole database."""
        nq = o_idx.shape[0]
        assert d_hits.is_contiguous() and d_hits.numel() * d_hits.element_size() == 16 * nq * nparts
        sp = C.c_void_p(stream) if stream else None
        self._before(stream)
        check(lib().orbx_knn2_merge_packed_device(self._m, int(nparts), nq, C.c_void_p(d_hits.data_ptr()), int(init),
                                                  C.c_void_p(o_idx.data_ptr()), C.c_void_p(o_d1.data_ptr()),
                                                  C.c_void_p(o_d2.data_ptr()), sp), "orbx_knn2_merge_packed_device")
        self._hold(stream, d_hits, o_idx, o_d1, o_d2)

    def after(self):
        """Makes torch's current stream wait (on the device) for everything queued on the matcher's stream so far."""
        import torch
        torch.cuda.current_stream().wait_event(self._own_stream().record_event())

    def merge_device(self, d_idx, d_d1, d_d2, o_idx, o_d1, o_d2, init=256, stream=None):
        """Merge (nparts, nq) partial results of consecutive database shards."""
        nparts, nq = d_idx.shape
        sp = C.c_void_p(stream) if stream else None
        self._before(stream)
        check(lib().orbx_knn2_merge_device(self._m, nparts, nq, C.c_void_p(d_idx.data_ptr()),
                                           C.c_void_p(d_d1.data_ptr()), C.c_void_p(d_d2.data_ptr()), int(init),
                                           C.c_void_p(o_idx.data_ptr()), C.c_void_p(o_d1.data_ptr()),
                                           C.c_void_p(o_d2.data_ptr()), sp), "orbx_knn2_merge_device")
        self._hold(stream, d_idx, d_d1, d_d2, o_idx, o_d1, o_d2)

    def ratio_test_device(self, d_d1, d_d2, d_accept, th=TH_LOW, inclusive=True, ratio=0.6, stream=None):
        sp = C.c_void_p(stream) if stream else None
        self._before(stream)
        check(lib().orbx_ratio_test_device(self._m, d_d1.shape[0], C.c_void_p(d_d1.data_ptr()),
                                           C.c_void_p(d_d2.data_ptr()), int(th), 1 if inclusive else 0,
                                           float(ratio), C.c_void_p(d_accept.data_ptr()), sp),
              "orbx_ratio_test_device")
        self._hold(stream, d_d1, d_d2, d_accept)

    def stereo_band_match_device(self, d_kl, d_dl, d_nl, d_kr, d_dr, d_nr, scale_factors, bf, b,
                                 d_best_idx, d_best_dist, stream=None):
        """Descriptor band search for (npairs, cap, ...) left/right keypoint sets on the device."""
        npairs, cap = d_dl.shape[0], d_dl.shape[1]
        sf = np.ascontiguousarray(scale_factors, dtype=np.float32)
        sp = C.c_void_p(stream) if stream else None
        self._before(stream)
        check(lib().orbx_stereo_band_match_device(
            self._m, npairs, cap, C.c_void_p(d_kl.data_ptr()), C.c_void_p(d_dl.data_ptr()),
            C.c_void_p(d_nl.data_ptr()), C.c_void_p(d_kr.data_ptr()), C.c_void_p(d_dr.data_ptr()),
            C.c_void_p(d_nr.data_ptr()), np_ptr(sf), sf.shape[0], float(bf), float(b),
            C.c_void_p(d_best_idx.data_ptr()), C.c_void_p(d_best_dist.data_ptr()), sp),
            "orbx_stereo_band_match_device")
        self._hold(stream, d_kl, d_dl, d_nl, d_kr, d_dr, d_nr, d_best_idx, d_best_dist)

    def compute_stereo_matches_device(self, left, right, d_kl, d_dl, d_nl, d_kr, d_dr, d_nr, bf, b,
                                      d_uright, d_depth, stream=None):
        """Frame::ComputeStereoMatches (src/Frame.cc:503-677) for the stereo pairs that `left` and `right`
        (two ORBextractor objects) extracted in their last batched call. Fills mvuRight / mvDepth."""
        npairs, cap = d_dl.shape[0], d_dl.shape[1]
        sp = C.c_void_p(stream) if stream else None
        self._before(stream)
        check(lib().orbx_compute_stereo_matches_device(
            self._m, left._ctx, right._ctx, npairs, cap, C.c_void_p(d_kl.data_ptr()), C.c_void_p(d_dl.data_ptr()),
            C.c_void_p(d_nl.data_ptr()), C.c_void_p(d_kr.data_ptr()), C.c_void_p(d_dr.data_ptr()),
            C.c_void_p(d_nr.data_ptr()), float(bf), float(b), C.c_void_p(d_uright.data_ptr()),
            C.c_void_p(d_depth.data_ptr()), sp), "orbx_compute_stereo_matches_device")
        self._hold(stream, d_kl, d_dl, d_nl, d_kr, d_dr, d_nr, d_uright, d_depth)

    def search_for_initialization(self, keys1_un, desc1, keys2_un, desc2, cell_start2, cell_items2, bounds,
                                  prev_matched, window_size=100, nnratio=0.9, check_orientation=True):
        """ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:413-530); the reference calls it with
        ORBmatcher(0.9, true) and windowSize 100 (src/Tracking.cc:670). Host numpy arrays.
        Returns (nmatches, matches12); prev_matched (n1, 2) float32 is updated in place."""
        from ._lib import KEYPOINT_DTYPE
        k1 = np.ascontiguousarray(keys1_un, dtype=KEYPOINT_DTYPE)
        k2 = np.ascontiguousarray(keys2_un, dtype=KEYPOINT_DTYPE)
        d1 = np.ascontiguousarray(desc1, dtype=np.uint8)
        d2 = np.ascontiguousarray(desc2, dtype=np.uint8)
        cs = np.ascontiguousarray(cell_start2, dtype=np.int32)
        ci = np.ascontiguousarray(cell_items2, dtype=np.int32)
        assert prev_matched.dtype == np.float32 and prev_matched.flags.c_contiguous and prev_matched.shape == (len(k1), 2)
        out = np.empty(len(k1), dtype=np.int32)
        n = C.c_int(0)
        check(lib().orbx_search_for_initialization(
            self._m, np_ptr(k1), np_ptr(d1), len(k1), np_ptr(k2), np_ptr(d2), len(k2), np_ptr(cs), np_ptr(ci),
            float(bounds[0]), float(bounds[1]), float(bounds[2]), float(bounds[3]), np_ptr(prev_matched),
            int(window_size), float(nnratio), 1 if check_orientation else 0, np_ptr(out), C.byref(n)),
            "orbx_search_for_initialization")
        return n.value, out

    def search_for_initialization_device(self, keys1_un, d_desc1, keys2_un, d_keys2_un, d_desc2, d_cell_start2,
                                         d_cell_items2, bounds, prev_matched, window_size=100, nnratio=0.9,
                                         check_orientation=True):
        """search_for_initialization for two frames whose descriptors, undistorted keypoints and grid are still on
        the GPU (torch CUDA tensors: d_desc1 (n1, 32) uint8, d_keys2_un (n2, 7) 28-byte records, d_desc2 (n2, 32),
        d_cell_start2 (3073,) int32, d_cell_items2 (>= n2,) int32: one frame's slices of what extract_batch_device
        and undistort_assign_grid_device wrote). keys1_un / keys2_un are the host copies of mvKeysUn."""
        from ._lib import KEYPOINT_DTYPE
        k1 = np.ascontiguousarray(keys1_un, dtype=KEYPOINT_DTYPE)
        k2 = np.ascontiguousarray(keys2_un, dtype=KEYPOINT_DTYPE)
        assert prev_matched.dtype == np.float32 and prev_matched.flags.c_contiguous and prev_matched.shape == (len(k1), 2)
        assert d_desc1.shape[0] >= len(k1) and d_desc2.shape[0] >= len(k2) and d_keys2_un.shape[0] >= len(k2)
        for t in (d_desc1, d_keys2_un, d_desc2, d_cell_start2, d_cell_items2):
            assert t.is_cuda and t.is_contiguous()
        out = np.empty(len(k1), dtype=np.int32)
        n = C.c_int(0)
        self._before(None)
        check(lib().orbx_search_for_initialization_device(
            self._m, np_ptr(k1), C.c_void_p(d_desc1.data_ptr()), len(k1), np_ptr(k2), C.c_void_p(d_keys2_un.data_ptr()),
            C.c_void_p(d_desc2.data_ptr()), len(k2), C.c_void_p(d_cell_start2.data_ptr()),
            C.c_void_p(d_cell_items2.data_ptr()), float(bounds[0]), float(bounds[1]), float(bounds[2]),
            float(bounds[3]), np_ptr(prev_matched), int(window_size), float(nnratio), 1 if check_orientation else 0,
            np_ptr(out), C.byref(n)), "orbx_search_for_initialization_device")
        return n.value, out

    def search_by_projection_frame(self, proj_uv, inv_z, last_octave, last_angle, desc_mp, has_observations,
                                   keys_un, desc, cell_start, cell_items, bounds, scale_factors, th, mbf=0.0,
                                   direction=0, u_right=None, occupied=None, check_orientation=True, max_dist=100):
        """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (src/ORBmatcher.cc:1372-1516) from the
        projection onwards; Tracking calls it with ORBmatcher(0.9, true) and th = 15 (7 for stereo), then 2 * th
        (src/Tracking.cc:929-940). Host numpy arrays; see include/orbx.h for the meaning of every argument.
        max_dist = 100 (TH_HIGH) for that overload; the relocalisation overload (:1518-1650) is the same call with
        max_dist = ORBdist (see include/orbx.h). Returns (nmatches, matches) with one entry per current-frame feature."""
        from ._lib import KEYPOINT_DTYPE
        uv = np.ascontiguousarray(proj_uv, dtype=np.float32).reshape(-1, 2)
        iz = np.ascontiguousarray(inv_z, dtype=np.float32)
        lo = np.ascontiguousarray(last_octave, dtype=np.int32)
        la = np.ascontiguousarray(last_angle, dtype=np.float32)
        dm = np.ascontiguousarray(desc_mp, dtype=np.uint8).reshape(-1, 32)
        ho = np.ascontiguousarray(has_observations, dtype=np.uint8)
        k2 = np.ascontiguousarray(keys_un, dtype=KEYPOINT_DTYPE)
        d2 = np.ascontiguousarray(desc, dtype=np.uint8)
        cs = np.ascontiguousarray(cell_start, dtype=np.int32)
        ci = np.ascontiguousarray(cell_items, dtype=np.int32)
        sf = np.ascontiguousarray(scale_factors, dtype=np.float32)
        ur = None if u_right is None else np.ascontiguousarray(u_right, dtype=np.float32)
        oc = None if occupied is None else np.ascontiguousarray(occupied, dtype=np.uint8)
        npts = len(uv)
        assert len(iz) == npts and len(lo) == npts and len(la) == npts and len(dm) == npts and len(ho) == npts
        assert ur is None or len(ur) == len(k2)
        assert oc is None or len(oc) == len(k2)
        out = np.empty(len(k2), dtype=np.int32)
        n = C.c_int(0)
        check(lib().orbx_search_by_projection_frame(
            self._m, np_ptr(uv), np_ptr(iz), np_ptr(lo), np_ptr(la), np_ptr(dm), np_ptr(ho), npts, np_ptr(k2),
            np_ptr(d2), None if ur is None else np_ptr(ur), None if oc is None else np_ptr(oc), len(k2), np_ptr(cs),
            np_ptr(ci), float(bounds[0]), float(bounds[1]), float(bounds[2]), float(bounds[3]), np_ptr(sf), len(sf),
            float(th), float(mbf), int(direction), int(max_dist), 1 if check_orientation else 0, np_ptr(out),
            C.byref(n)), "orbx_search_by_projection_frame")
        return n.value, out

    def search_by_projection_points(self, proj_xy, view_cos, level, desc_mp, has_observations, keys_un, desc,
                                    cell_start, cell_items, bounds, scale_factors, th=1.0, nnratio=0.8, proj_xr=None,
                                    u_right=None, occupied=None):
        """ORBmatcher::SearchByProjection(F, vpMapPoints, th) (src/ORBmatcher.cc:41-123); Tracking::SearchLocalPoints
        calls it with ORBmatcher(0.8) and th = 1, 3 or 5 (src/Tracking.cc:1253-1263). Host numpy arrays; see
        include/orbx.h. Returns (nmatches, matches) with one entry per feature of F."""
        from ._lib import KEYPOINT_DTYPE
        xy = np.ascontiguousarray(proj_xy, dtype=np.float32).reshape(-1, 2)
        vc = np.ascontiguousarray(view_cos, dtype=np.float32)
        lv = np.ascontiguousarray(level, dtype=np.int32)
        dm = np.ascontiguousarray(desc_mp, dtype=np.uint8).reshape(-1, 32)
        ho = np.ascontiguousarray(has_observations, dtype=np.uint8)
        k2 = np.ascontiguousarray(keys_un, dtype=KEYPOINT_DTYPE)
        d2 = np.ascontiguousarray(desc, dtype=np.uint8)
        cs = np.ascontiguousarray(cell_start, dtype=np.int32)
        ci = np.ascontiguousarray(cell_items, dtype=np.int32)
        sf = np.ascontiguousarray(scale_factors, dtype=np.float32)
        xr = None if proj_xr is None else np.ascontiguousarray(proj_xr, dtype=np.float32)
        ur = None if u_right is None else np.ascontiguousarray(u_right, dtype=np.float32)
        oc = None if occupied is None else np.ascontiguousarray(occupied, dtype=np.uint8)
        npts = len(xy)
        assert len(vc) == npts and len(lv) == npts and len(dm) == npts and len(ho) == npts
        assert (ur is None or (len(ur) == len(k2) and xr is not None and len(xr) == npts))
        assert oc is None or len(oc) == len(k2)
        out = np.empty(len(k2), dtype=np.int32)
        n = C.c_int(0)
        check(lib().orbx_search_by_projection_points(
            self._m, np_ptr(xy), None if xr is None else np_ptr(xr), np_ptr(vc), np_ptr(lv), np_ptr(dm), np_ptr(ho),
            npts, np_ptr(k2), np_ptr(d2), None if ur is None else np_ptr(ur), None if oc is None else np_ptr(oc),
            len(k2), np_ptr(cs), np_ptr(ci), float(bounds[0]), float(bounds[1]), float(bounds[2]), float(bounds[3]),
            np_ptr(sf), len(sf), float(th), float(nnratio), np_ptr(out), C.byref(n)),
            "orbx_search_by_projection_points")
        return n.value, out

    def search_by_bow(self, keys1_un, desc1, valid1, fv1, keys2, desc2, fv2, valid2=None, nnratio=0.7,
                      check_orientation=True):
        """ORBmatcher::SearchByBoW. valid2 None: the KeyFrame x Frame overload (src/ORBmatcher.cc:153-293,
        called with ORBmatcher(0.7, true) at src/Tracking.cc:840-843); matches has one entry per side-2
        feature. valid2 given: the KeyFrame x KeyFrame overload (:532-690); one entry per side-1 feature.
        fvK = (nodes, start, items): ascending node ids, features of node j = items[start[j]:start[j+1]].
        Returns (nmatches, matches)."""
        from ._lib import KEYPOINT_DTYPE
        k1 = np.ascontiguousarray(keys1_un, dtype=KEYPOINT_DTYPE)
        k2 = np.ascontiguousarray(keys2, dtype=KEYPOINT_DTYPE)
        d1 = np.ascontiguousarray(desc1, dtype=np.uint8)
        d2 = np.ascontiguousarray(desc2, dtype=np.uint8)
        v1 = np.ascontiguousarray(valid1, dtype=np.uint8)
        v2 = None if valid2 is None else np.ascontiguousarray(valid2, dtype=np.uint8)
        a = [np.ascontiguousarray(fv1[0], dtype=np.uint32), np.ascontiguousarray(fv1[1], dtype=np.int32),
             np.ascontiguousarray(fv1[2], dtype=np.uint32)]
        b = [np.ascontiguousarray(fv2[0], dtype=np.uint32), np.ascontiguousarray(fv2[1], dtype=np.int32),
             np.ascontiguousarray(fv2[2], dtype=np.uint32)]
        mode = 0 if v2 is None else 1
        out = np.empty(len(k2) if mode == 0 else len(k1), dtype=np.int32)
        n = C.c_int(0)
        check(lib().orbx_search_by_bow(
            self._m, mode, np_ptr(k1), np_ptr(d1), np_ptr(v1), len(k1), np_ptr(a[0]), np_ptr(a[1]), np_ptr(a[2]),
            len(a[0]), np_ptr(k2), np_ptr(d2), None if v2 is None else np_ptr(v2), len(k2), np_ptr(b[0]),
            np_ptr(b[1]), np_ptr(b[2]), len(b[0]), float(nnratio), 1 if check_orientation else 0, np_ptr(out),
            C.byref(n)), "orbx_search_by_bow")
        return n.value, out

    def search_by_bow_device(self, keys1_un, d_desc1, valid1, fv1, keys2, d_desc2, fv2, valid2=None, nnratio=0.7,
                             check_orientation=True, prepared=None):
        """search_by_bow with the two descriptor arrays on the GPU (torch CUDA uint8 tensors (n, 32), e.g. one
        frame's slice of what extract_batch_device wrote); everything else as in search_by_bow."""
        from ._lib import KEYPOINT_DTYPE
        k1 = np.ascontiguousarray(keys1_un, dtype=KEYPOINT_DTYPE)
        k2 = np.ascontiguousarray(keys2, dtype=KEYPOINT_DTYPE)
        v1 = np.ascontiguousarray(valid1, dtype=np.uint8)
        v2 = None if valid2 is None else np.ascontiguousarray(valid2, dtype=np.uint8)
        a = [np.ascontiguousarray(fv1[0], dtype=np.uint32), np.ascontiguousarray(fv1[1], dtype=np.int32),
             np.ascontiguousarray(fv1[2], dtype=np.uint32)]
        b = [np.ascontiguousarray(fv2[0], dtype=np.uint32), np.ascontiguousarray(fv2[1], dtype=np.int32),
             np.ascontiguousarray(fv2[2], dtype=np.uint32)]
        assert d_desc1.is_cuda and d_desc2.is_cuda and d_desc1.is_contiguous() and d_desc2.is_contiguous()
        assert d_desc1.shape[0] >= len(k1) and d_desc2.shape[0] >= len(k2)
        mode = 0 if v2 is None else 1
        out = np.empty(len(k2) if mode == 0 else len(k1), dtype=np.int32)
        n = C.c_int(0)
        self._before(None)
        check(lib().orbx_search_by_bow_device(
            self._m, mode, np_ptr(k1), C.c_void_p(d_desc1.data_ptr()), np_ptr(v1), len(k1), np_ptr(a[0]), np_ptr(a[1]),
            np_ptr(a[2]), len(a[0]), np_ptr(k2), C.c_void_p(d_desc2.data_ptr()), None if v2 is None else np_ptr(v2),
            len(k2), np_ptr(b[0]), np_ptr(b[1]), np_ptr(b[2]), len(b[0]), float(nnratio),
            1 if check_orientation else 0, np_ptr(out), C.byref(n)), "orbx_search_by_bow_device")
        return n.value, out

    @staticmethod
    def image_bounds(cam, width, height):
        """Frame::ComputeImageBounds (src/Frame.cc:474-501) -> (minX, minY, maxX, maxY)."""
        out = [C.c_float(0) for _ in range(4)]
        check(lib().orbx_image_bounds(C.byref(cam), width, height, *[C.byref(o) for o in out]), "orbx_image_bounds")
        return tuple(o.value for o in out)

    def undistort_assign_grid_device(self, d_kps, d_counts, cam, bounds, d_kps_un, d_cell_start, d_cell_items,
                                     stream=None):
        """Frame::UndistortKeyPoints + AssignFeaturesToGrid for (nframes, cap, 7) keypoint tensors.
        d_cell_start is (nframes, 64*48+1) int32, d_cell_items (nframes, cap) int32."""
        nframes, cap = d_kps.shape[0], d_kps.shape[1]
        sp = C.c_void_p(stream) if stream else None
        self._before(stream)
        check(lib().orbx_undistort_assign_grid_device(
            self._m, nframes, cap, C.c_void_p(d_kps.data_ptr()), C.c_void_p(d_counts.data_ptr()), C.byref(cam),
            float(bounds[0]), float(bounds[1]), float(bounds[2]), float(bounds[3]), C.c_void_p(d_kps_un.data_ptr()),
            C.c_void_p(d_cell_start.data_ptr()), C.c_void_p(d_cell_items.data_ptr()), sp),
            "orbx_undistort_assign_grid_device")
        self._hold(stream, d_kps, d_counts, d_kps_un, d_cell_start, d_cell_items)

    def synchronize(self):
        check(lib().orbx_matcher_synchronize(self._m), "orbx_matcher_synchronize")
        self._inflight.clear()

    def _own_stream(self):
        import torch
        if self._ext_stream is None:
            h = C.c_void_p()
            check(lib().orbx_matcher_get_stream(self._m, C.byref(h)), "orbx_matcher_get_stream")
            self._ext_stream = torch.cuda.ExternalStream(h.value, device=torch.device("cuda", self._device))
        return self._ext_stream

    def _before(self, stream):
        """The matcher's stream does not synchronise with torch's streams by itself: let it wait for what torch
        has queued so far on the current stream (producers of the inputs, fills of the outputs)."""
        if stream:
            return
        import torch
        self._own_stream().wait_event(torch.cuda.current_stream().record_event())

    def _hold(self, stream, *tensors):
        """The device calls return before their kernels run, on a stream torch's allocator does not know.
        Keeps the tensors referenced until an event recorded behind the call has completed."""
        if stream:
            return
        import torch
        ev = torch.cuda.Event()
        ev.record(self._own_stream())
        self._inflight.append((ev, tensors))
        while self._inflight and self._inflight[0][0].query():
            self._inflight.popleft()
