"""MI355X-native ORB front end: drop-in for ORB-SLAM2's ORBextractor and Hamming matcher.

The compute path is liborbx.so (hand-written gfx950 HIP kernels behind the C ABI of
include/orbx.h).  This package is the thin Python host mirror of the reference's C++ interface
(my_orb_slam2/include/ORBextractor.h, ORBmatcher.h); it never computes anything itself and raises
if the library is missing.
"""
from ._lib import (KEYPOINT_DTYPE, BLUR_CV_4_0_0, BLUR_CV_4_2_PLUS, Camera, OrbxError, lib, lib_path)
from .extractor import ORBextractor
from .matcher import ORBmatcher, HammingMatcher, TH_HIGH, TH_LOW, HISTO_LENGTH
from .synth import synth_frame, synth_batch

__all__ = [
    "KEYPOINT_DTYPE", "BLUR_CV_4_0_0", "BLUR_CV_4_2_PLUS", "Camera", "OrbxError", "lib", "lib_path",
    "ORBextractor", "ORBmatcher", "HammingMatcher", "TH_HIGH", "TH_LOW", "HISTO_LENGTH",
    "synth_frame", "synth_batch",
]
