#!/usr/bin/env python3
"""Timing of the steps either side of extraction, the GPU path beside the CPU oracle on the same inputs:

  undistort + grid      Frame::UndistortKeyPoints + AssignFeaturesToGrid, 256 frames of ~1000 keypoints, on the device
  ComputeStereoMatches  Frame::ComputeStereoMatches on 128 extracted pairs (nFeatures 1200), on the device
  SearchForInitialization   two 2000-keypoint frames (the initialisation extractor), host arrays in and out
  SearchByProjection    (CurrentFrame, LastFrame): ~700 map points into a 1000-keypoint frame, th 15
  SearchByBoW           both overloads, 1000 x 1200 features over 120 vocabulary nodes and 2000 x 2000 over 40,
                        host arrays in and out

    python tests/campaigns/bench_next_rows.py

One JSON object per step. The oracle is single-threaded, as the reference runs these on the tracking thread.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def best_of(fn, reps, inner=1):
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(inner):
            fn()
        best = min(best, (time.perf_counter() - t0) / inner)
    return best


def main():
    import torch
    import my_orb_slam2_amd as m
    from tests import oracle as orc
    from tests.test_gpu_frame_grid import EUROC
    from tests.test_gpu_search_bow import make_case

    # ---- undistort + grid ----
    B = 256
    ex = m.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
    cap = ex.capacity
    d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex.extract_batch_device(torch.from_numpy(m.synth_batch(0, B)).cuda(), d_kps, d_desc, d_cnt)
    ex.synchronize()
    hm = m.HammingMatcher()
    cam = m.Camera(**EUROC)
    bounds = hm.image_bounds(cam, 752, 480)
    d_un = torch.zeros_like(d_kps)
    d_start = torch.zeros((B, 64 * 48 + 1), dtype=torch.int32, device="cuda")
    d_items = torch.zeros((B, cap), dtype=torch.int32, device="cuda")

    def gpu_grid():
        hm.undistort_assign_grid_device(d_kps, d_cnt, cam, bounds, d_un, d_start, d_items)
        hm.synchronize()
    gpu_grid()
    t_gpu = best_of(gpu_grid, 5, 4)
    cnt = d_cnt.cpu().numpy()
    kps = d_kps.cpu().numpy().view(np.uint8).reshape(B, cap, 28).copy().view(m.KEYPOINT_DTYPE).reshape(B, cap)
    K4 = [EUROC[k] for k in ("fx", "fy", "cx", "cy")]
    D5 = [EUROC[k] for k in ("k1", "k2", "p1", "p2", "k3")]

    def cpu_grid():
        for f in range(16):
            k = kps[f, :cnt[f]].copy()
            xy = orc.undistort_points(np.stack([k["x"], k["y"]], 1), K4, D5)
            k["x"], k["y"] = xy[:, 0], xy[:, 1]
            orc.assign_grid(k, bounds)
    t_cpu = best_of(cpu_grid, 3) / 16
    print(json.dumps({"step": "undistort + grid", "frames": B, "mean_keypoints": float(cnt.mean()),
                      "gpu_us_per_frame": round(t_gpu / B * 1e6, 3), "gpu_frames_per_s": round(B / t_gpu),
                      "cpu_port_us_per_frame": round(t_cpu * 1e6, 1), "cpu_cores": 1}))

    # ---- ComputeStereoMatches (extraction not included on either side) ----
    n = 128
    left, right = m.synth_batch(5000, n), m.synth_batch(5000, n, right_view=True)
    exl = m.ORBextractor(1200, 1.2, 8, 20, 7, max_batch=n)
    exr = m.ORBextractor(1200, 1.2, 8, 20, 7, max_batch=n)
    scap = exl.capacity
    outs = [(torch.zeros((n, scap, 7), dtype=torch.float32, device="cuda"),
             torch.zeros((n, scap, 32), dtype=torch.uint8, device="cuda"),
             torch.zeros(n, dtype=torch.int32, device="cuda")) for _ in range(2)]
    d_left, d_right = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    exl.extract_batch_device(d_left, *outs[0])
    exr.extract_batch_device(d_right, *outs[1])
    exl.synchronize()
    exr.synchronize()
    bf, b = 47.90639384423901, 47.90639384423901 / 435.2046959714599
    ur = torch.empty((n, scap), dtype=torch.float32, device="cuda")
    z = torch.empty((n, scap), dtype=torch.float32, device="cuda")

    def gpu_stereo():
        hm.compute_stereo_matches_device(exl, exr, *outs[0], *outs[1], bf, b, ur, z)
        hm.synchronize()
    gpu_stereo()
    t_gpu = best_of(gpu_stereo, 5, 4)
    ol, orr = orc.Extractor(1200), orc.Extractor(1200)
    t_cpu = 0.0
    for p in range(4):
        okl, odl = ol.extract(left[p])
        okr, odr = orr.extract(right[p])
        t_cpu += best_of(lambda: orc.compute_stereo_matches(ol, orr, okl, odl, okr, odr, bf, b), 3) / 4
    print(json.dumps({"step": "ComputeStereoMatches", "pairs": n, "matched_per_pair": float((ur >= 0).sum().item()) / n,
                      "gpu_us_per_pair": round(t_gpu / n * 1e6, 3), "gpu_pairs_per_s": round(n / t_gpu),
                      "cpu_port_us_per_pair": round(t_cpu * 1e6, 1), "cpu_cores": 1}))
    del exl, exr

    # ---- SearchForInitialization ----
    ex2 = m.ORBextractor(2000, 1.2, 8, 20, 7)
    k1, d1 = ex2(m.synth_frame(80))
    k2, d2 = ex2(m.synth_frame(80, right_view=True))
    b0 = (0.0, 0.0, 752.0, 480.0)
    start, items = orc.assign_grid(k2, b0)
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
    n_gpu, _ = hm.search_for_initialization(k1, d1, k2, d2, start, items, b0, prev.copy())
    t_gpu = best_of(lambda: hm.search_for_initialization(k1, d1, k2, d2, start, items, b0, prev.copy()), 5, 4)
    t_cpu = best_of(lambda: orc.search_for_initialization(k1, d1, k2, d2, b0, prev.copy(), 100, 0.9, True), 3)
    # the same search on device-resident frames: descriptors, mvKeysUn and the grid stay in HBM
    import ctypes as C
    from my_orb_slam2_amd._lib import lib, np_ptr
    t_d1, t_d2 = torch.from_numpy(d1).cuda(), torch.from_numpy(d2).cuda()
    t_k2 = torch.from_numpy(k2.view(np.uint8).reshape(-1, 28).copy()).cuda()
    t_start, t_items = torch.from_numpy(start).cuda(), torch.from_numpy(np.ascontiguousarray(items, np.int32)).cuda()
    torch.cuda.synchronize()
    out12 = np.empty(len(k1), np.int32)
    nm = C.c_int(0)
    pv = prev.copy()

    init_args = (hm._m, np_ptr(k1), C.c_void_p(t_d1.data_ptr()), len(k1), np_ptr(k2), C.c_void_p(t_k2.data_ptr()),
                 C.c_void_p(t_d2.data_ptr()), len(k2), C.c_void_p(t_start.data_ptr()), C.c_void_p(t_items.data_ptr()),
                 0.0, 0.0, 752.0, 480.0, np_ptr(pv), 100, 0.9, 1, np_ptr(out12), C.byref(nm))
    f_init = lib().orbx_search_for_initialization_device

    def init_dev():
        pv[:] = prev
        rc = f_init(*init_args)
        assert rc == 0
    init_dev()
    assert nm.value == n_gpu
    t_dev = best_of(init_dev, 7, 8)
    print(json.dumps({"step": "SearchForInitialization", "n1": len(k1), "n2": len(k2), "matches": int(n_gpu),
                      "gpu_path_ms": round(t_gpu * 1e3, 3), "gpu_device_resident_ms": round(t_dev * 1e3, 3),
                      "cpu_port_ms": round(t_cpu * 1e3, 3), "cpu_cores": 1,
                      "note": "gpu_path: host arrays in and out through the Python binding; gpu_device_resident: "
                              "orbx_search_for_initialization_device called through the C ABI on data left in HBM; "
                              "both include the host-side greedy pass"}))

    # ---- SearchByProjection(CurrentFrame, LastFrame) ----
    from tests.test_gpu_search_projection import make_case as proj_case, BOUNDS
    c = proj_case(1, 15, 0, False)
    pa = (c["uv"], c["inv_z"], c["last_octave"], c["last_angle"], c["desc_mp"], c["has_obs"])
    n_gpu, _ = hm.search_by_projection_frame(*pa, c["kc"], c["dc"], c["start"], c["items"], BOUNDS, c["sf"], c["th"])
    t_gpu = best_of(lambda: hm.search_by_projection_frame(*pa, c["kc"], c["dc"], c["start"], c["items"], BOUNDS,
                                                          c["sf"], c["th"]), 5, 8)
    t_cpu = best_of(lambda: orc.search_by_projection_frame(*pa, c["kc"], c["dc"], BOUNDS, c["sf"], c["th"]), 3, 2)
    print(json.dumps({"step": "SearchByProjection(Frame, LastFrame)", "map_points": len(c["uv"]), "n": len(c["kc"]),
                      "th": c["th"], "matches": int(n_gpu), "gpu_path_ms": round(t_gpu * 1e3, 3),
                      "cpu_port_ms": round(t_cpu * 1e3, 3), "cpu_cores": 1,
                      "note": "host arrays in and out; includes the copies and the host-side greedy pass"}))

    # ---- SearchByBoW ----
    for case in ((1, 1000, 1200, 120), (2, 2000, 2000, 40)):
        k1, d1, v1, fv1, k2, d2, v2, fv2 = make_case(*case)
        for name, kw in (("SearchByBoW(KeyFrame, Frame)", {}), ("SearchByBoW(KeyFrame, KeyFrame)", {"valid2": v2})):
            n_gpu, _ = hm.search_by_bow(k1, d1, v1, fv1, k2, d2, fv2, **kw)
            t_gpu = best_of(lambda: hm.search_by_bow(k1, d1, v1, fv1, k2, d2, fv2, **kw), 5, 8)
            t_cpu = best_of(lambda: orc.search_by_bow(k1, d1, v1, fv1, k2, d2, fv2, **kw), 3, 2)
            # both forms through the C ABI with arguments made once (the Python binding's own conversions cost
            # tens of microseconds, comparable to the call)
            t_d1, t_d2 = torch.from_numpy(d1).cuda(), torch.from_numpy(d2).cuda()
            torch.cuda.synchronize()
            a = [np.ascontiguousarray(fv1[0], np.uint32), np.ascontiguousarray(fv1[1], np.int32),
                 np.ascontiguousarray(fv1[2], np.uint32)]
            bb = [np.ascontiguousarray(fv2[0], np.uint32), np.ascontiguousarray(fv2[1], np.int32),
                  np.ascontiguousarray(fv2[2], np.uint32)]
            mode = 1 if kw else 0
            outm = np.empty(len(k1) if mode else len(k2), np.int32)
            nmb = C.c_int(0)
            v2p = np_ptr(v2) if kw else None

            def args(p1, p2):
                return (hm._m, mode, np_ptr(k1), p1, np_ptr(v1), len(k1), np_ptr(a[0]), np_ptr(a[1]), np_ptr(a[2]),
                        len(a[0]), np_ptr(k2), p2, v2p, len(k2), np_ptr(bb[0]), np_ptr(bb[1]), np_ptr(bb[2]), len(bb[0]),
                        0.7, 1, np_ptr(outm), C.byref(nmb))
            host_args = args(np_ptr(d1), np_ptr(d2))
            dev_args = args(C.c_void_p(t_d1.data_ptr()), C.c_void_p(t_d2.data_ptr()))
            f_host, f_dev = lib().orbx_search_by_bow, lib().orbx_search_by_bow_device

            def bow_c(fn, arg):
                rc = fn(*arg)
                assert rc == 0 and nmb.value == n_gpu
            host = lambda: bow_c(f_host, host_args)
            dev = lambda: bow_c(f_dev, dev_args)
            host(); dev()
            t_host_c = best_of(host, 7, 16)
            t_dev_c = best_of(dev, 7, 16)
            print(json.dumps({"step": name, "n1": len(k1), "n2": len(k2), "nodes": len(fv1[0]), "matches": int(n_gpu),
                              "gpu_path_ms": round(t_gpu * 1e3, 3), "gpu_host_arrays_c_abi_ms": round(t_host_c * 1e3, 4),
                              "gpu_device_resident_ms": round(t_dev_c * 1e3, 4), "cpu_port_ms": round(t_cpu * 1e3, 3),
                              "cpu_cores": 1,
                              "note": "gpu_path: host arrays through the Python binding; *_c_abi / device_resident: the "
                                      "C entry points called with ready-made arguments; all include the host-side "
                                      "greedy pass"}))

if __name__ == "__main__":
    main()
