#!/usr/bin/env python3
"""Randomised parity of the steps either side of extraction: kNN top-2, ComputeStereoMatches,
UndistortKeyPoints + AssignFeaturesToGrid, SearchForInitialization, SearchByBoW, each against the CPU oracle with random
sizes, settings, cameras and windows. Exits 1 with the failing case on the first difference.

    python tests/campaigns/fuzz_match.py [--cases 60] [--seed 1]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def fail(what, params):
    print("FAIL:", what, json.dumps(params))
    sys.exit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import torch
    import my_orb_slam2_amd as m
    from tests import oracle as orc
    rng = np.random.default_rng(a.seed)
    hm = m.HammingMatcher()
    done = {"knn": 0, "stereo_pairs": 0, "stereo_matches": 0, "grid_frames": 0, "init_matches": 0, "projection_matches": 0, "local_map_matches": 0, "bow_matches": 0}

    for case in range(a.cases):
        # ---- kNN ----
        nq, ndb = int(rng.integers(1, 700)), int(rng.integers(0, 6000))
        db = rng.integers(0, 256, (ndb, 32), dtype=np.uint8)
        if ndb > 10:
            db[rng.integers(0, ndb, ndb // 5)] = db[rng.integers(0, ndb, ndb // 5)]          # duplicates
        q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        if ndb:
            q[: nq // 2] = db[rng.integers(0, ndb, nq // 2)]
            q[: nq // 4, 0] ^= 3
        init = int(rng.choice([256, 100, 2**31 - 1]))
        got, want = hm.knn2(q, db, init=init), orc.knn2(q, db, init=init)
        if not all(np.array_equal(g, w) for g, w in zip(got, want)):
            fail("knn2", dict(case=case, nq=nq, ndb=ndb, init=init))
        done["knn"] += 1

        # ---- stereo ----
        w, h = [(752, 480), (640, 480), (321, 243), (1241, 376)][int(rng.integers(0, 4))]
        nf = int(rng.choice([200, 1200, 2000]))
        npairs = int(rng.integers(1, 4))
        idx0 = int(rng.integers(0, 100000))
        left = m.synth_batch(idx0, npairs, w, h)
        right = m.synth_batch(idx0, npairs, w, h, right_view=True)
        if rng.random() < 0.2:
            right[-1] = 0                                                                       # a blank right image
        exl = m.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=npairs)
        exr = m.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=npairs)
        cap = exl.capacity

        def alloc():
            return (torch.zeros((npairs, cap, 7), dtype=torch.float32, device="cuda"),
                    torch.zeros((npairs, cap, 32), dtype=torch.uint8, device="cuda"),
                    torch.zeros(npairs, dtype=torch.int32, device="cuda"))
        kl, dl, nl = alloc()
        kr, dr, nr = alloc()
        exl.extract_batch_device(torch.from_numpy(left).cuda(), kl, dl, nl)
        exr.extract_batch_device(torch.from_numpy(right).cuda(), kr, dr, nr)
        fx = float(rng.uniform(300, 600))
        bf = float(rng.uniform(20, 80))
        b = bf / fx
        ur = torch.empty((npairs, cap), dtype=torch.float32, device="cuda")
        z = torch.empty((npairs, cap), dtype=torch.float32, device="cuda")
        hm.compute_stereo_matches_device(exl, exr, kl, dl, nl, kr, dr, nr, bf, b, ur, z)
        hm.synchronize()
        urh, zh, nlh = ur.cpu().numpy(), z.cpu().numpy(), nl.cpu().numpy()
        for p in range(npairs):
            ol, orr = orc.Extractor(nf), orc.Extractor(nf)
            okl, odl = ol.extract(left[p])
            okr, odr = orr.extract(right[p])
            our, oz, kept = orc.compute_stereo_matches(ol, orr, okl, odl, okr, odr, bf, b)
            if nlh[p] != len(okl) or not np.array_equal(urh[p, :nlh[p]], our) or not np.array_equal(zh[p, :nlh[p]], oz):
                why = "count %d vs %d" % (nlh[p], len(okl)) if nlh[p] != len(okl) else "uRight differs at %d of %d, depth at %d" % (
                    int((urh[p, :nlh[p]] != our).sum()), nlh[p], int((zh[p, :nlh[p]] != oz).sum()))
                fail("stereo: " + why, dict(case=case, w=w, h=h, nf=nf, npairs=npairs, pair=p, idx0=idx0, bf=bf, b=b,
                                            blank_right=bool((right[-1] == 0).all())))
            done["stereo_pairs"] += 1
            done["stereo_matches"] += kept

        # ---- undistort + grid on the left keypoints ----
        cam_kw = dict(fx=fx, fy=float(rng.uniform(300, 600)), cx=w / 2 + float(rng.uniform(-20, 20)),
                      cy=h / 2 + float(rng.uniform(-20, 20)), k1=float(rng.uniform(-0.35, 0.1)), k2=float(rng.uniform(-0.05, 0.1)),
                      p1=float(rng.uniform(-1e-3, 1e-3)), p2=float(rng.uniform(-1e-3, 1e-3)), k3=float(rng.uniform(-0.01, 0.01)))
        if rng.random() < 0.25:
            cam_kw.update(k1=0.0)
        cam = m.Camera(**cam_kw)
        K4 = [cam_kw["fx"], cam_kw["fy"], cam_kw["cx"], cam_kw["cy"]]
        D5 = [cam_kw["k1"], cam_kw["k2"], cam_kw["p1"], cam_kw["p2"], cam_kw["k3"]]
        bounds = hm.image_bounds(cam, w, h)
        if not (bounds[2] > bounds[0] and bounds[3] > bounds[1]):     # a distortion that folds the image: no grid
            cam_kw.update(k1=0.0, k2=0.0, p1=0.0, p2=0.0, k3=0.0)
            cam = m.Camera(**cam_kw)
            D5 = [0.0] * 5
            bounds = hm.image_bounds(cam, w, h)
        d_un = torch.zeros_like(kl)
        d_start = torch.zeros((npairs, 64 * 48 + 1), dtype=torch.int32, device="cuda")
        d_items = torch.full((npairs, cap), -1, dtype=torch.int32, device="cuda")
        hm.undistort_assign_grid_device(kl, nl, cam, bounds, d_un, d_start, d_items)
        hm.synchronize()
        view = lambda t: t.cpu().numpy().view(np.uint8).reshape(npairs, cap, 28).copy().view(m.KEYPOINT_DTYPE).reshape(npairs, cap)
        kps, un = view(kl), view(d_un)
        start, items = d_start.cpu().numpy(), d_items.cpu().numpy()
        for f in range(npairs):
            k = kps[f, :nlh[f]]
            want = k.copy()
            if cam_kw["k1"] != 0 and len(k):
                xy = orc.undistort_points(np.stack([k["x"], k["y"]], 1), K4, D5)
                want["x"], want["y"] = xy[:, 0], xy[:, 1]
            ostart, oitems = orc.assign_grid(want, bounds)
            if not (np.array_equal(un[f, :nlh[f]], want) and np.array_equal(start[f], ostart) and
                    np.array_equal(items[f, :ostart[-1]], oitems)):
                fail("undistort_grid", dict(case=case, w=w, h=h, nf=nf, idx=idx0 + f, cam=cam_kw))
            done["grid_frames"] += 1

        # ---- SearchForInitialization between the left and right view of pair 0 ----
        view_r = view(kr)
        nrh = nr.cpu().numpy()
        k1, k2 = kps[0, :nlh[0]].copy(), view_r[0, :nrh[0]].copy()
        d1, d2 = dl[0, :nlh[0]].cpu().numpy(), dr[0, :nrh[0]].cpu().numpy()
        ib = (0.0, 0.0, float(w), float(h))
        s2, i2 = orc.assign_grid(k2, ib)
        prev = (np.stack([k1["x"], k1["y"]], 1) + rng.uniform(-20, 20, (len(k1), 2))).astype(np.float32)
        window = int(rng.choice([10, 50, 100, 200]))
        ratio = float(rng.choice([0.6, 0.9]))
        ori = bool(rng.integers(0, 2))
        p_o = prev.copy()
        want_n, want, want_prev = orc.search_for_initialization(k1, d1, k2, d2, ib, p_o, window, ratio, ori)
        p_g = prev.copy()
        n, got = hm.search_for_initialization(k1, d1, k2, d2, s2, i2 if len(i2) else np.zeros(1, np.int32), ib, p_g,
                                              window, ratio, ori)
        if n != want_n or not np.array_equal(got, want) or not np.array_equal(p_g, want_prev):
            fail("search_for_initialization", dict(case=case, w=w, h=h, nf=nf, idx=idx0, window=window, ratio=ratio, ori=ori))
        done["init_matches"] += n
        # the same search on device-resident descriptors, keypoints and grid
        if len(k1) and len(k2):
            tt = lambda arr: torch.from_numpy(np.ascontiguousarray(arr)).cuda()
            p_d = prev.copy()
            n, got = hm.search_for_initialization_device(k1, tt(d1), k2, tt(k2.view(np.uint8).reshape(-1, 28)), tt(d2),
                                                         tt(s2), tt(i2.astype(np.int32) if len(i2) else np.zeros(1, np.int32)),
                                                         ib, p_d, window, ratio, ori)
            if n != want_n or not np.array_equal(got, want) or not np.array_equal(p_d, want_prev):
                fail("search_for_initialization_device", dict(case=case, w=w, h=h, nf=nf, idx=idx0, window=window,
                                                              ratio=ratio, ori=ori))

        # ---- SearchByBoW, both overloads: a feature's vocabulary node is the top bits of its first descriptor byte ----
        if len(k1) and len(k2):
            shift = int(rng.integers(1, 7))

            def feature_vector(d):
                node = (d[:, 0] >> shift).astype(np.uint32) * 3 + 1
                ids = np.unique(node)
                order = [rng.permutation(np.flatnonzero(node == nid)) for nid in ids]
                starts = np.concatenate([[0], np.cumsum([len(o) for o in order])]).astype(np.int32)
                return ids.astype(np.uint32), starts, np.concatenate(order).astype(np.uint32)
            fv1, fv2 = feature_vector(d1), feature_vector(d2)
            v1 = (rng.random(len(k1)) < 0.8).astype(np.uint8)
            v2 = (rng.random(len(k2)) < 0.8).astype(np.uint8)
            for valid2 in (None, v2):
                want_n, want = orc.search_by_bow(k1, d1, v1, fv1, k2, d2, fv2, valid2, ratio, ori)
                n, got = hm.search_by_bow(k1, d1, v1, fv1, k2, d2, fv2, valid2, ratio, ori)
                n_d, got_d = hm.search_by_bow_device(k1, tt(d1), v1, fv1, k2, tt(d2), fv2, valid2, ratio, ori)
                if n != want_n or not np.array_equal(got, want) or n_d != want_n or not np.array_equal(got_d, want):
                    fail("search_by_bow", dict(case=case, w=w, h=h, nf=nf, idx=idx0, shift=shift, ratio=ratio, ori=ori,
                                               kfkf=valid2 is not None))
                done["bow_matches"] += n

        # ---- SearchByProjection(current = right view, last = left view): random flow, depths, occupancy ----
        if len(k1) and len(k2):
            pick = np.flatnonzero(rng.random(len(k1)) < rng.uniform(0.2, 1.0))
            npts = len(pick)
            uv = (np.stack([k1["x"][pick], k1["y"][pick]], 1) + rng.normal(0, float(rng.uniform(0.5, 8)), (npts, 2))
                  ).astype(np.float32)
            uv[:, 0] -= rng.uniform(0, 40, npts).astype(np.float32)
            inv_z = rng.uniform(0.01, 1.0, npts).astype(np.float32)
            inv_z[rng.random(npts) < 0.05] *= -1
            uv[rng.random(npts) < 0.05] += np.float32(w)
            dmp = d1[pick].copy()
            flips = rng.integers(0, 256, (npts, 32), dtype=np.uint8) & rng.integers(0, 256, (npts, 32), dtype=np.uint8)
            dmp ^= flips & np.uint8(rng.choice([0x01, 0x11, 0x55]))
            has_obs = (rng.random(npts) < 0.8).astype(np.uint8)
            occ = (rng.random(len(k2)) < 0.15).astype(np.uint8) if rng.random() < 0.7 else None
            stereo = rng.random() < 0.5
            ur2 = np.where(rng.random(len(k2)) < 0.6, k2["x"] - rng.uniform(0, 40, len(k2)), -1).astype(np.float32) if stereo else None
            th = float(rng.choice([7, 14, 15, 30]))
            direction = int(rng.integers(0, 3)) if stereo else 0
            sfac = exl.GetScaleFactors()
            pargs = (uv, inv_z, k1["octave"][pick].astype(np.int32), k1["angle"][pick].astype(np.float32), dmp, has_obs)
            want_n, want = orc.search_by_projection_frame(*pargs, k2, d2, ib, sfac, th, bf, direction, ur2, occ, ori)
            n, got = hm.search_by_projection_frame(*pargs, k2, d2, s2, i2 if len(i2) else np.zeros(1, np.int32), ib, sfac,
                                                   th, bf, direction, ur2, occ, ori)
            if n != want_n or not np.array_equal(got, want):
                fail("search_by_projection_frame", dict(case=case, w=w, h=h, nf=nf, idx=idx0, th=th, direction=direction,
                                                        stereo=stereo, ori=ori, n=n, want_n=want_n))
            done["projection_matches"] += n
            # the local-map overload on the same data
            vcos = rng.uniform(0.99, 1.0, npts).astype(np.float32)
            lvl = np.clip(k1["octave"][pick] + rng.integers(-1, 2, npts), 0, 7).astype(np.int32)
            xr = (uv[:, 0] - rng.uniform(0, 40, npts)).astype(np.float32) if stereo else None
            th2 = float(rng.choice([1.0, 3.0, 5.0]))
            want_n, want = orc.search_by_projection_points(uv, vcos, lvl, dmp, has_obs, k2, d2, ib, sfac, th2, ratio, xr, ur2, occ)
            n, got = hm.search_by_projection_points(uv, vcos, lvl, dmp, has_obs, k2, d2, s2,
                                                    i2 if len(i2) else np.zeros(1, np.int32), ib, sfac, th2, ratio, xr, ur2, occ)
            if n != want_n or not np.array_equal(got, want):
                fail("search_by_projection_points", dict(case=case, w=w, h=h, nf=nf, idx=idx0, th=th2, ratio=ratio, stereo=stereo))
            done["local_map_matches"] += n
        exl.close()
        exr.close()
    print(json.dumps(dict(cases=a.cases, seed=a.seed, differences=0, **done)))


if __name__ == "__main__":
    main()
