#!/usr/bin/env python3
"""Randomised parity: random image sizes, extractor settings and frame kinds, GPU against the CPU oracle.

Every case builds a fresh context, extracts one frame (sometimes through an unaligned view) and compares all
keypoint fields, angles (1e-3 deg) and descriptors. Prints the failing case's parameters and exits 1 on the
first difference.

    python tests/campaigns/fuzz_parity.py [--cases 300] [--seed 1]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--reuse", action="store_true",
                    help="one context sized 1000 x 800 for all cases (image sizes change from call to call; batches of 1-3, "
                         "which take the one-launch pyramid, and every fourth case 9-12, which go level by level)")
    a = ap.parse_args()
    import my_orb_slam2_amd as m
    from my_orb_slam2_amd import synth as S
    from tests import oracle as orc
    rng = np.random.default_rng(a.seed)
    total_kp = 0
    if a.reuse:
        gpu = m.ORBextractor(700, 1.2, 8, 20, 7, max_width=1000, max_height=800, max_batch=12)
        cpu = orc.Extractor(700, 1.2, 8, 20, 7)
        for case in range(a.cases):
            w, h, b = int(rng.integers(70, 1001)), int(rng.integers(70, 801)), int(rng.integers(1, 4))
            if case % 4 == 3:
                b = int(rng.integers(9, 13))
            frames = m.synth_batch(5000 + 3 * case, b, w, h)
            if rng.random() < 0.2:
                frames[0] = m.synth_frame(case, w, h, kind=S.NOISE)
            got = gpu.extract_batch(frames) if b > 1 or rng.random() < 0.5 else [gpu(frames[0])]
            for i, (gk, gd) in enumerate(got):
                ck, cd = cpu.extract(frames[i], fast=True)
                ok = len(gk) == len(ck) and all(np.array_equal(gk[f], ck[f]) for f in ("x", "y", "size", "response", "octave", "angle"))
                if not (ok and np.array_equal(gd, cd)):
                    print("FAIL (reused context):", json.dumps(dict(case=case, w=w, h=h, batch=b, frame=i)), len(gk), len(ck))
                    sys.exit(1)
                total_kp += len(ck)
        print(json.dumps({"cases": a.cases, "seed": a.seed, "mode": "one context, changing sizes", "keypoints_compared": total_kp,
                          "differences": 0}))
        return
    for case in range(a.cases):
        w = int(rng.integers(70, 1000))
        h = int(rng.integers(70, 800))
        if rng.random() < 0.15:
            w, h = int(rng.integers(600, 2000)), int(rng.integers(70, 140))       # panoramas: many root nodes
        kw = dict(nfeatures=int(rng.choice([5, 20, 60, 300, 1000, 2500])), scale=float(rng.choice([1.05, 1.1, 1.2, 1.33, 1.5, 2.0])),
                  nlevels=int(rng.integers(1, 13)), ini=int(rng.integers(5, 60)))
        kw["mn"] = int(rng.integers(2, kw["ini"] + 1))
        blur = int(rng.integers(0, 2))
        kind = int(rng.choice([S.SCENE, S.SCENE, S.SCENE, S.NOISE, S.CONST128]))
        params = dict(case=case, w=w, h=h, blur=blur, kind=kind, **kw)
        try:
            gpu = m.ORBextractor(kw["nfeatures"], kw["scale"], kw["nlevels"], kw["ini"], kw["mn"], max_width=w, max_height=h,
                                 blur_taps=blur)
        except m.OrbxError as e:            # settings the library refuses (say so and move on)
            print("refused:", json.dumps(params), str(e)[:80])
            continue
        cpu = orc.Extractor(kw["nfeatures"], kw["scale"], kw["nlevels"], kw["ini"], kw["mn"], blur)
        img = m.synth_frame(1000 + case, w, h, kind=kind)
        if rng.random() < 0.3:              # odd base address and pitch
            big = np.zeros((h + 1, w + 7), np.uint8)
            big[1:, 3:3 + w] = img
            img = big[1:, 3:3 + w]
        ck, cd = cpu.extract(np.ascontiguousarray(img), fast=True)
        try:
            gk, gd = gpu(img)
        except m.OrbxError as e:
            print("FAIL (error):", json.dumps(params), str(e))
            sys.exit(1)
        ok = len(gk) == len(ck)
        if ok and len(ck):
            ok = all(np.array_equal(gk[f], ck[f]) for f in ("x", "y", "size", "response", "octave", "class_id"))
            da = np.abs(gk["angle"] - ck["angle"])
            ok = ok and float(np.minimum(da, 360 - da).max()) <= 1e-3 and np.array_equal(gd, cd)
        if not ok:
            print("FAIL:", json.dumps(params), "gpu", len(gk), "cpu", len(ck))
            sys.exit(1)
        total_kp += len(ck)
        gpu.close()
    print(json.dumps({"cases": a.cases, "seed": a.seed, "keypoints_compared": total_kp, "differences": 0}))


if __name__ == "__main__":
    main()
