#!/usr/bin/env python3
"""Writes tests/golden/*.npz: regression vectors produced by the CPU oracle (oracle/) on seeded
synthetic frames.  They are NOT outputs of the reference (which cannot be built here and ships no
fixtures); they pin today's oracle so that a later change to it, or to the frame generator, is
noticed, and they let GPU tests compare against stored data as well as against the live oracle.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import my_orb_slam2_amd as m  # noqa: E402
from tests import oracle as orc  # noqa: E402

out = os.path.dirname(os.path.abspath(__file__))
# extraction: frame 0 at 752x480 / 1000 features, and a small odd-sized frame
for name, idx, w, h, nf in [("extract_752x480_f0", 0, 752, 480, 1000), ("extract_321x243_f11", 11, 321, 243, 500)]:
    e = orc.Extractor(nf, 1.2, 8, 20, 7, orc.BLUR_CV_4_0_0)
    k, d = e.extract(m.synth_frame(idx, w, h))
    np.savez_compressed(os.path.join(out, name + ".npz"), keypoints=k, descriptors=d,
                        level_sizes=np.array([e.level_size(l) for l in range(8)]),
                        candidates_per_level=np.array([len(e.level_candidates(l)) for l in range(8)]),
                        level3_crc=np.array([int(np.frombuffer(e.level_image(3).tobytes(), np.uint8).astype(np.uint64).sum())]))
    print(name, len(k))
# matcher: planted queries
rng = np.random.default_rng(2026)
db = rng.integers(0, 256, (4096, 32), dtype=np.uint8)
q = rng.integers(0, 256, (256, 32), dtype=np.uint8)
q[:128] = db[rng.integers(0, 4096, 128)]
q[:128, 5] ^= 0x11
idx, d1, d2 = orc.knn2(q, db)
np.savez_compressed(os.path.join(out, "knn2_256x4096.npz"), query=q, db=db, idx=idx, d1=d1, d2=d2)
print("knn2", int((d1 <= 50).sum()))
