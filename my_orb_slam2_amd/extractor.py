"""Host mirror of ORB_SLAM2::ORBextractor (my_orb_slam2/include/ORBextractor.h:46-126).

Same constructor arguments, same getters, same call: extractor(image, mask) -> keypoints,
descriptors.  All computation happens in liborbx.so on the GPU.
"""
import collections
import ctypes as C

import numpy as np

from ._lib import (BLUR_CV_4_0_0, KEYPOINT_DTYPE, Config, check, lib, np_ptr)


class ORBextractor:
    HARRIS_SCORE = 0  # include/ORBextractor.h:50
    FAST_SCORE = 1

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, *, device=0,
                 max_width=752, max_height=480, max_batch=1, blur_taps=BLUR_CV_4_0_0):
        """ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
        (include/ORBextractor.h:52-53).  The keyword arguments size the device context; blur_taps
        selects the Gaussian tap set (OpenCV 4.0.0, which the reference pins, by default)."""
        self._ctx = C.c_void_p()
        cfg = Config(int(nfeatures), float(scaleFactor), int(nlevels), int(iniThFAST), int(minThFAST),
                     int(device), int(max_width), int(max_height), int(max_batch), int(blur_taps))
        check(lib().orbx_create(C.byref(cfg), C.byref(self._ctx)), "orbx_create")
        self.nfeatures = int(nfeatures)
        self.nlevels = int(nlevels)
        self.max_batch = int(max_batch)
        self.device = int(device)
        self._capacity = lib().orbx_keypoint_capacity(self._ctx)
        self._last_shape = None
        self._ext_stream = None
        self._inflight = collections.deque()
        self._call_bufs = None

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            if self._inflight:
                lib().orbx_synchronize(self._ctx)
                self._inflight.clear()
            self._ext_stream = None
            lib().orbx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- getters (include/ORBextractor.h:70-98) ----
    def GetLevels(self):
        return lib().orbx_get_levels(self._ctx)

    def GetScaleFactor(self):
        return lib().orbx_get_scale_factor(self._ctx)

    def _floats(self, fn):
        out = np.empty(self.nlevels, dtype=np.float32)
        check(fn(self._ctx, np_ptr(out)), fn.__name__)
        return out

    def GetScaleFactors(self):
        return self._floats(lib().orbx_get_scale_factors)

    def GetInverseScaleFactors(self):
        return self._floats(lib().orbx_get_inverse_scale_factors)

    def GetScaleSigmaSquares(self):
        return self._floats(lib().orbx_get_scale_sigma_squares)

    def GetInverseScaleSigmaSquares(self):
        return self._floats(lib().orbx_get_inverse_scale_sigma_squares)

    def features_per_level(self):
        out = np.empty(self.nlevels, dtype=np.int32)
        check(lib().orbx_get_features_per_level(self._ctx, np_ptr(out)), "orbx_get_features_per_level")
        return out

    def umax(self):
        out = np.empty(16, dtype=np.int32)
        check(lib().orbx_get_umax(self._ctx, np_ptr(out)), "orbx_get_umax")
        return out

    @property
    def capacity(self):
        """Keypoint capacity that no image the context accepts can overflow (orbx_keypoint_capacity)."""
        return self._capacity

    # ---- operator() (src/ORBextractor.cc:1035-1097) ----
    def __call__(self, image, mask=None):
        """Returns (keypoints, descriptors): a KEYPOINT_DTYPE array of N records and an (N, 32)
        uint8 array.  `mask` is ignored, as in the reference (include/ORBextractor.h:65).  An empty
        image returns empty outputs (:1038-1039)."""
        if image is None or image.size == 0:
            return np.empty(0, dtype=KEYPOINT_DTYPE), np.empty((0, 32), dtype=np.uint8)
        if image.dtype != np.uint8 or image.ndim != 2:
            raise TypeError("image must be a 2-D uint8 array (CV_8UC1, src/ORBextractor.cc:1042)")
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        h, w = image.shape
        if self._call_bufs is None:
            # output arrays and their pointers are made once: `a.ctypes` costs microseconds, which matter when the
            # call itself takes a tenth of a millisecond
            kps = np.empty(self._capacity, dtype=KEYPOINT_DTYPE)
            desc = np.empty((self._capacity, 32), dtype=np.uint8)
            n = C.c_int(0)
            self._call_bufs = (kps, desc, n, np_ptr(kps), np_ptr(desc), C.byref(n), lib().orbx_extract)
        kps, desc, n, pk, pd, pn, fn = self._call_bufs
        n.value = 0
        check(fn(self._ctx, image.ctypes.data, w, h, image.strides[0], pk, pd, self._capacity, pn), "orbx_extract")
        self._last_shape = (1, h, w)
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch(self, images):
        """Host batch: images is (B, H, W) uint8.  Returns a list of (keypoints, descriptors)."""
        images = np.ascontiguousarray(images, dtype=np.uint8)
        b, h, w = images.shape
        cap = self._capacity
        kps = np.empty((b, cap), dtype=KEYPOINT_DTYPE)
        desc = np.empty((b, cap, 32), dtype=np.uint8)
        counts = np.empty(b, dtype=np.int32)
        check(lib().orbx_extract_batch(self._ctx, np_ptr(images), b, w, h, images.strides[1], images.strides[0],
                                       np_ptr(kps), np_ptr(desc), cap, np_ptr(counts)), "orbx_extract_batch")
        self._last_shape = (b, h, w)
        if int(counts.max(initial=0)) > cap:
            raise RuntimeError("orbx_extract_batch: %d keypoints in a frame, capacity %d" % (int(counts.max()), cap))
        return [(kps[i, :counts[i]].copy(), desc[i, :counts[i]].copy()) for i in range(b)]

    def extract_batch_device(self, d_images, d_keypoints, d_descriptors, d_counts, stream=None):
        """Device batch on torch CUDA tensors (no copies, no host synchronisation; call synchronize() or order
        your own stream against the context's before reading the outputs):
        d_images (B, H, W) uint8, d_keypoints (B, cap, 7) float32/int32 view (28-byte records),
        d_descriptors (B, cap, 32) uint8, d_counts (B,) int32."""
        b, h, w = d_images.shape
        if d_images.stride(2) != 1:
            raise ValueError("frames must have unit pixel stride")
        # The kernels read level 0 in place with dword loads over whole padded rows: the last frame is read up to
        # h * row_stride bytes, not just to its last pixel. A view whose storage ends earlier (a narrower window into
        # a larger image that ends with the allocation) is copied to a dense tensor first.
        need = (b - 1) * d_images.stride(0) + h * d_images.stride(1)
        avail = d_images.untyped_storage().nbytes() - d_images.storage_offset() * d_images.element_size()
        if avail < need:
            d_images = d_images.contiguous()
        cap = d_descriptors.shape[1]
        assert d_keypoints.shape[0] == b and d_keypoints.shape[1] == cap and d_keypoints.element_size() * \
            d_keypoints.shape[2] == 28 and d_keypoints.is_contiguous() and d_descriptors.is_contiguous()
        sp = C.c_void_p(stream) if stream else None
        if not stream:
            # the context's stream does not synchronise with torch's streams by itself: let it wait for whatever
            # torch has queued so far on the current stream (the producers of d_images, fills of the outputs)
            import torch
            self._torch_stream().wait_event(torch.cuda.current_stream().record_event())
        check(lib().orbx_extract_batch_device(self._ctx, C.c_void_p(d_images.data_ptr()), b, w, h,
                                              d_images.stride(1), d_images.stride(0),
                                              C.c_void_p(d_keypoints.data_ptr()),
                                              C.c_void_p(d_descriptors.data_ptr()), cap,
                                              C.c_void_p(d_counts.data_ptr()), sp), "orbx_extract_batch_device")
        self._last_shape = (b, h, w)
        if not stream:
            # The work runs on the context's stream, which torch's allocator does not know: a tensor dropped
            # right after this call could be handed out again while the kernels still use it. Keep the
            # tensors referenced until an event recorded behind the call has completed.
            import torch
            ev = torch.cuda.Event()
            ev.record(self._torch_stream())
            self._inflight.append((ev, (d_images, d_keypoints, d_descriptors, d_counts)))
            while self._inflight and self._inflight[0][0].query():
                self._inflight.popleft()

    def _torch_stream(self):
        if self._ext_stream is None:
            import torch
            h = C.c_void_p()
            check(lib().orbx_get_stream(self._ctx, C.byref(h)), "orbx_get_stream")
            self._ext_stream = torch.cuda.ExternalStream(h.value, device=torch.device("cuda", self.device))
        return self._ext_stream

    def synchronize(self):
        check(lib().orbx_synchronize(self._ctx), "orbx_synchronize")
        self._inflight.clear()

    STAGES = ("pyramid", "fast_cells", "octree", "orient_desc")

    def set_profiling(self, enabled=True):
        check(lib().orbx_set_profiling(self._ctx, 1 if enabled else 0), "orbx_set_profiling")

    def stage_ms(self):
        """Milliseconds the four stages of the last profiled batched call took on the GPU."""
        out = np.empty(4, dtype=np.float32)
        check(lib().orbx_get_stage_ms(self._ctx, np_ptr(out)), "orbx_get_stage_ms")
        return dict(zip(self.STAGES, out.tolist()))

    # ---- mvImagePyramid (include/ORBextractor.h:100) and stage probes ----
    def level_size(self, level):
        w, h = C.c_int(0), C.c_int(0)
        check(lib().orbx_level_size(self._ctx, level, C.byref(w), C.byref(h)), "orbx_level_size")
        return w.value, h.value

    def get_level(self, level, frame=0, border=False):
        w, h = self.level_size(level)
        pad = 38 if border else 0
        out = np.empty((h + pad, w + pad), dtype=np.uint8)
        check(lib().orbx_get_level(self._ctx, frame, level, 1 if border else 0, np_ptr(out), out.strides[0]),
              "orbx_get_level")
        return out

    @property
    def mvImagePyramid(self):
        """Levels of frame 0 of the last call, without the 19-px border frame."""
        return [self.get_level(l) for l in range(self.nlevels)]

    BLUR_PATCH_SHAPE = (37, 44)      # ORBX_BLUR_PATCH_ROWS x ORBX_BLUR_PATCH_PITCH
    BLUR_PATCH_CENTRE = (18, 23)     # row, column of the keypoint inside its patch

    def debug_keypoint_blur(self, frame=0):
        """Blurred patch of every keypoint of `frame` of the last call, (N, 37, 44) uint8, as the descriptor
        stage of the kernel read it (only offsets the rotated pattern can reach are defined)."""
        n = C.c_int(0)
        check(lib().orbx_debug_keypoint_blur(self._ctx, frame, None, 0, C.byref(n)), "orbx_debug_keypoint_blur")
        out = np.zeros((max(n.value, 1),) + self.BLUR_PATCH_SHAPE, dtype=np.uint8)
        check(lib().orbx_debug_keypoint_blur(self._ctx, frame, np_ptr(out), out.shape[0], C.byref(n)),
              "orbx_debug_keypoint_blur")
        return out[:n.value]

    def _probe(self, fn, level, frame):
        n = C.c_int(0)
        check(fn(self._ctx, frame, level, None, 0, C.byref(n)), fn.__name__)
        out = np.empty(max(n.value, 1), dtype=KEYPOINT_DTYPE)
        check(fn(self._ctx, frame, level, np_ptr(out), out.shape[0], C.byref(n)), fn.__name__)
        return out[:n.value]

    def debug_candidates(self, level, frame=0):
        return self._probe(lib().orbx_debug_candidates, level, frame)

    def debug_selected(self, level, frame=0):
        return self._probe(lib().orbx_debug_selected, level, frame)
