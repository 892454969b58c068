"""CPU tests of the host side: the C-ABI library loads, exports every declared symbol, and fails
loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import hashlib
import os
import re

import numpy as np
import pytest

import my_orb_slam2_amd as m
from my_orb_slam2_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "orbx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(orbx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 30
    handle = C.CDLL(m.lib_path())
    for n in names:
        assert hasattr(handle, n), n
    assert sorted(_lib.SIGNATURES) == names          # the Python binding covers the whole header


def test_keypoint_layout_is_cv_keypoint():
    assert m.KEYPOINT_DTYPE.itemsize == 28
    assert [m.KEYPOINT_DTYPE.fields[f][1] for f in ("x", "y", "size", "angle", "response", "octave", "class_id")] \
        == [0, 4, 8, 12, 16, 20, 24]


def test_synth_frames_are_deterministic():
    a = m.synth_frame(3)
    b = m.synth_frame(3)
    assert np.array_equal(a, b) and a.shape == (480, 752)
    assert not np.array_equal(a, m.synth_frame(4))
    golden = open(os.path.join(ROOT, "tests", "golden", "synth_sha256.txt")).read().split()
    got = [hashlib.sha256(m.synth_frame(i).tobytes()).hexdigest() for i in range(2)]
    got.append(hashlib.sha256(m.synth_frame(0, right_view=True).tobytes()).hexdigest())
    got.append(hashlib.sha256(m.synth_frame(0, 1920, 1080).tobytes()).hexdigest())
    assert got == golden
    assert m.synth_frame(0, kind=m.synth.ZERO).max() == 0
    assert np.all(m.synth_frame(0, kind=m.synth.CONST128) == 128)


def test_scalar_distance_on_host():
    z = np.zeros(32, np.uint8)
    assert m.ORBmatcher.DescriptorDistance(z, z) == 0
    assert m.ORBmatcher.DescriptorDistance(z, np.full(32, 255, np.uint8)) == 256
    assert (m.TH_HIGH, m.TH_LOW, m.HISTO_LENGTH) == (100, 50, 30)


def test_no_gpu_means_error_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(m.OrbxError) as e:
        m.ORBextractor(1000, 1.2, 8, 20, 7)
    assert e.value.code == -2                            # ORBX_ERR_HIP
    with pytest.raises(m.OrbxError):
        m.HammingMatcher()


def test_one_launch_pyramid_plan_covers_every_pixel_once():
    """Host logic behind the pyramid kernel of small calls: for many image sizes, pyramid steps, level counts and
    tile shapes the plan must give every pixel of every level to exactly one tile and keep every tile's source
    pixels inside its own tile of the level above (checked inside the library, which needs no GPU for it)."""
    from my_orb_slam2_amd._lib import lib
    rng = np.random.default_rng(5)
    cases = [(752, 480, 1.2, 8, 16, 8, 8), (1920, 1080, 1.2, 8, 16, 8, 8), (1241, 376, 1.2, 8, 16, 8, 8),
             (752, 480, 1.2, 8, 64, 32, 4), (333, 250, 1.1, 12, 16, 8, 8), (500, 300, 1.7, 4, 16, 8, 8),
             (97, 80, 1.2, 8, 16, 8, 8), (4096, 4096, 1.2, 3, 16, 8, 8), (752, 480, 2.0, 4, 16, 8, 8)]
    for _ in range(400):
        cases.append((int(rng.integers(40, 2000)), int(rng.integers(40, 1300)),
                      float(rng.choice([1.05, 1.1, 1.2, 1.25, 1.33, 1.5, 1.7, 2.0])), int(rng.integers(2, 13)),
                      int(rng.choice([8, 16, 24, 32, 64])), int(rng.choice([1, 4, 8, 16, 32])), int(rng.integers(1, 9))))
    planned = 0
    for w, h, scale, nlevels, tw, th, gl in cases:
        n = C.c_int(-1)
        rc = lib().orbx_debug_check_pyramid_plan(w, h, scale, nlevels, tw, th, gl, C.byref(n))
        assert rc == 0, (w, h, scale, nlevels, tw, th, gl, lib().orbx_last_error())
        planned += n.value > 0
    assert planned > 300          # most shapes take the one-launch kernel; the rest fall back by design
    n = C.c_int(-1)
    assert lib().orbx_debug_check_pyramid_plan(752, 480, 1.2, 8, 16, 8, 8, C.byref(n)) == 0 and n.value == 1
    assert lib().orbx_debug_check_pyramid_plan(333, 250, 1.1, 12, 16, 8, 8, C.byref(n)) == 0 and n.value == 2
