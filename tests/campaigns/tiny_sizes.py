#!/usr/bin/env python3
"""Images too small for the extractor's borders and cells (1 x 1 up to 80 x 80, all eight levels requested):
the GPU path must neither fault nor differ from the oracle, which yields no keypoints for levels the reference
itself cannot process (it divides by a zero cell count there, src/ORBextractor.cc:776-779).

    python tests/campaigns/tiny_sizes.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    import my_orb_slam2_amd as m
    from tests import oracle as orc
    rng = np.random.default_rng(5)
    sizes = [(1, 1), (1, 80), (80, 1), (2, 2), (7, 7), (16, 16), (31, 31), (32, 32), (33, 33), (37, 37), (38, 38),
             (39, 39), (45, 45), (50, 38), (38, 50), (61, 61), (62, 62), (63, 63), (64, 64), (70, 45), (45, 70)]
    sizes += [(int(rng.integers(1, 81)), int(rng.integers(1, 81))) for _ in range(150)]
    gpu = m.ORBextractor(300, 1.2, 8, 20, 7, max_width=80, max_height=80)
    cpu = orc.Extractor(300, 1.2, 8, 20, 7)
    bad = total = 0
    for w, h in sizes:
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        gk, gd = gpu(img)
        ck, cd = cpu.extract(img)
        total += len(ck)
        if len(gk) != len(ck) or not np.array_equal(gd, cd) or any(
                not np.array_equal(gk[f], ck[f]) for f in ("x", "y", "size", "response", "octave")):
            bad += 1
            print("difference at %dx%d: %d vs %d keypoints" % (w, h, len(gk), len(ck)))
    print(json.dumps({"sizes": len(sizes), "keypoints_compared": total, "differences": bad}))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
