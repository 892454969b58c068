"""Seeded synthetic frames (liborbx's orbx_synth_frame): identical bytes on every machine."""
import numpy as np

from ._lib import check, lib, np_ptr

SCENE, ZERO, CONST128, NOISE = 0, 1, 2, 3


def synth_frame(index, width=752, height=480, kind=SCENE, nrect=None, ncheck=None, right_view=False):
    """Frame `index` as a (height, width) uint8 array. Object counts default to 300 rectangles and
    40 checker patches per 752x480 of area."""
    area = (width * height) / float(752 * 480)
    if nrect is None:
        nrect = int(round(300 * area))
    if ncheck is None:
        ncheck = int(round(40 * area))
    out = np.empty((height, width), dtype=np.uint8)
    check(lib().orbx_synth_frame(int(index), width, height, int(kind), int(nrect), int(ncheck),
                                 1 if right_view else 0, np_ptr(out), out.strides[0]), "orbx_synth_frame")
    return out


def synth_batch(start, count, width=752, height=480, right_view=False):
    out = np.empty((count, height, width), dtype=np.uint8)
    for i in range(count):
        out[i] = synth_frame(start + i, width, height, right_view=right_view)
    return out
