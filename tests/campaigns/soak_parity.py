#!/usr/bin/env python3
"""Parity at BASELINE.json's full sizes: the GPU path against the CPU oracle on every frame.

  configs[1]  4096 synthetic 752x480 frames (+ the three degenerate kinds), nFeatures 1000
  configs[3]  --hd N: N synthetic 1920x1080 frames, nFeatures 2000
  configs[2]  --stereo N: N synthetic stereo pairs, nFeatures 1200, whole ComputeStereoMatches
  configs[4]  --knn NQ NDB: top-2 Hamming search, half the queries planted with up to 40 flipped bits

Every keypoint field is compared exactly, angles within 1e-3 deg, descriptors bit for bit; stereo uRight /
depth exactly. Prints one JSON object; exits 1 on any difference. The oracle runs on host threads.

    python tests/campaigns/soak_parity.py [--frames 4096] [--hd 64] [--stereo 128] [--threads 16]
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def compare(gk, gd, ck, cd, stats):
    stats["frames"] += 1
    if len(gk) != len(ck):
        stats["count_mismatch_frames"] += 1
        return
    stats["keypoints"] += len(ck)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        stats["field_mismatches"] += int((gk[f] != ck[f]).sum())
    if len(ck):
        da = np.abs(gk["angle"] - ck["angle"])
        da = np.minimum(da, 360 - da)
        stats["max_angle_err_deg"] = max(stats["max_angle_err_deg"], float(da.max()))
        stats["angle_over_tol"] += int((da > 1e-3).sum())
        stats["descriptor_bits"] += cd.size * 8
        stats["descriptor_bit_mismatches"] += int(np.unpackbits(gd ^ cd).sum())


def run_extract(m, orc, torch, frames, nfeat, threads, batch, stats):
    h, w = frames.shape[1:]
    ex = m.ORBextractor(nfeat, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=batch)
    cap = ex.capacity
    pool = ThreadPoolExecutor(threads)
    oracles = [orc.Extractor(nfeat) for _ in range(threads)]

    # one oracle per worker slot: frames i, i + threads, ... go to the same extractor, so run in strides
    def cpu_stride(t):
        return [(i, oracles[t].extract(frames[i], fast=True)) for i in range(t, len(frames), threads)]
    futs = [pool.submit(cpu_stride, t) for t in range(threads)]
    gk_all, gd_all = [], []
    for b0 in range(0, len(frames), batch):
        chunk = frames[b0:b0 + batch]
        d = torch.from_numpy(chunk).cuda()
        k = torch.zeros((len(chunk), cap, 7), dtype=torch.float32, device="cuda")
        ds = torch.zeros((len(chunk), cap, 32), dtype=torch.uint8, device="cuda")
        c = torch.zeros(len(chunk), dtype=torch.int32, device="cuda")
        ex.extract_batch_device(d, k, ds, c)
        ex.synchronize()
        kh = k.cpu().numpy().view(m.KEYPOINT_DTYPE).reshape(len(chunk), cap)
        dh, ch = ds.cpu().numpy(), c.cpu().numpy()
        for i in range(len(chunk)):
            gk_all.append(kh[i, :ch[i]])
            gd_all.append(dh[i, :ch[i]])
    for f in futs:
        for i, (ck, cd) in f.result():
            compare(gk_all[i], gd_all[i], ck, cd, stats)
    pool.shutdown()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--hd", type=int, default=64)
    ap.add_argument("--stereo", type=int, default=128)
    ap.add_argument("--knn", type=int, nargs=2, default=[16384, 4 * 1024 * 1024], metavar=("NQ", "NDB"),
                    help="configs[4] scaled: planted + random queries against NDB rows, top-2 and ratio test")
    ap.add_argument("--threads", type=int, default=16)
    a = ap.parse_args()
    import torch
    import my_orb_slam2_amd as m
    from tests import oracle as orc
    out = {}
    t0 = time.time()

    def fresh():
        return {"frames": 0, "keypoints": 0, "count_mismatch_frames": 0, "field_mismatches": 0, "angle_over_tol": 0,
                "max_angle_err_deg": 0.0, "descriptor_bits": 0, "descriptor_bit_mismatches": 0}

    if a.frames:
        st = fresh()
        frames = m.synth_batch(0, a.frames, 752, 480)
        from my_orb_slam2_amd import synth as S
        for slot, kind in enumerate((S.ZERO, S.CONST128, S.NOISE)):
            frames[-1 - slot] = m.synth_frame(7, 752, 480, kind=kind)
        run_extract(m, orc, torch, frames, 1000, a.threads, 256, st)
        out["configs[1] 752x480 nFeatures 1000"] = st
    if a.hd:
        st = fresh()
        frames = m.synth_batch(0, a.hd, 1920, 1080)
        run_extract(m, orc, torch, frames, 2000, a.threads, 32, st)
        out["configs[3] 1920x1080 nFeatures 2000"] = st
    if a.stereo:
        n = a.stereo
        left = m.synth_batch(5000, n, 752, 480)
        right = m.synth_batch(5000, n, 752, 480, right_view=True)
        exl = m.ORBextractor(1200, 1.2, 8, 20, 7, max_batch=n)
        exr = m.ORBextractor(1200, 1.2, 8, 20, 7, max_batch=n)
        cap = exl.capacity

        def alloc():
            return (torch.zeros((n, cap, 7), dtype=torch.float32, device="cuda"),
                    torch.zeros((n, cap, 32), dtype=torch.uint8, device="cuda"),
                    torch.zeros(n, dtype=torch.int32, device="cuda"))
        kl, dl, nl = alloc()
        kr, dr, nr = alloc()
        d_left, d_right = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()   # kept alive: calls are async
        exl.extract_batch_device(d_left, kl, dl, nl)
        exr.extract_batch_device(d_right, kr, dr, nr)
        exl.synchronize()
        exr.synchronize()
        bf, b = 47.90639384423901, 47.90639384423901 / 435.2046959714599
        ur = torch.empty((n, cap), dtype=torch.float32, device="cuda")
        z = torch.empty((n, cap), dtype=torch.float32, device="cuda")
        hm = m.HammingMatcher()
        hm.compute_stereo_matches_device(exl, exr, kl, dl, nl, kr, dr, nr, bf, b, ur, z)
        hm.synchronize()
        ur, z, nlh = ur.cpu().numpy(), z.cpu().numpy(), nl.cpu().numpy()

        def cpu_pair(p):
            ol, orr = orc.Extractor(1200), orc.Extractor(1200)
            okl, odl = ol.extract(left[p])
            okr, odr = orr.extract(right[p])
            our, oz, kept = orc.compute_stereo_matches(ol, orr, okl, odl, okr, odr, bf, b)
            return p, len(okl), our, oz, kept
        st = {"pairs": 0, "left_keypoints": 0, "stereo_matches": 0, "count_mismatch_pairs": 0, "uright_mismatches": 0,
              "depth_mismatches": 0}
        with ThreadPoolExecutor(a.threads) as pool:
            for p, nk, our, oz, kept in pool.map(cpu_pair, range(n)):
                st["pairs"] += 1
                if nk != nlh[p]:
                    st["count_mismatch_pairs"] += 1
                    continue
                st["left_keypoints"] += nk
                st["stereo_matches"] += kept
                st["uright_mismatches"] += int((ur[p, :nk] != our).sum())
                st["depth_mismatches"] += int((z[p, :nk] != oz).sum())
        out["configs[2] stereo pairs nFeatures 1200"] = st
    if a.knn[0] and a.knn[1]:
        nq, ndb = a.knn
        rng = np.random.default_rng(1)
        db = rng.integers(0, 256, (ndb, 32), dtype=np.uint8)
        q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        half = nq // 2
        src = rng.integers(0, ndb, half)
        q[:half] = db[src]
        flips = rng.integers(0, 41, half)                       # r in [0, 40] random bits flipped
        for i in range(half):
            for bit in rng.integers(0, 256, flips[i]):
                q[i, bit >> 3] ^= np.uint8(1 << (bit & 7))
        hm = m.HammingMatcher()
        idx, d1, d2 = hm.knn2(q, db)
        acc = np.array([orc.lib().orc_ratio_accept(int(x), int(y), 50, 1, 0.6) for x, y in zip(d1, d2)], dtype=np.uint8)
        step = (nq + a.threads - 1) // a.threads
        with ThreadPoolExecutor(a.threads) as pool:
            parts = list(pool.map(lambda s0: orc.knn2(q[s0:s0 + step], db, 256, True), range(0, nq, step)))
        oi = np.concatenate([p[0] for p in parts])
        o1 = np.concatenate([p[1] for p in parts])
        o2 = np.concatenate([p[2] for p in parts])
        out["configs[4] scaled kNN"] = {"queries": nq, "database_rows": ndb, "pairs": nq * ndb,
                                        "idx_mismatches": int((idx != oi).sum()), "d1_mismatches": int((d1 != o1).sum()),
                                        "d2_mismatches": int((d2 != o2).sum()), "accepted_by_ratio_test": int(acc.sum())}
    out["seconds"] = round(time.time() - t0, 1)
    print(json.dumps(out, indent=1))
    bad = 0
    for v in out.values():
        if isinstance(v, dict):
            bad += sum(v.get(k, 0) for k in ("count_mismatch_frames", "field_mismatches", "angle_over_tol",
                                             "descriptor_bit_mismatches", "count_mismatch_pairs", "uright_mismatches",
                                             "depth_mismatches", "idx_mismatches", "d1_mismatches", "d2_mismatches"))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
