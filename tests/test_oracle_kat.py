"""CPU tests: the oracle against analytic known answers and against independent numpy
restatements.  The reference ships no golden vectors (parity unpinned), so these are the pins."""
import numpy as np
import pytest

from tests import oracle as orc


# ---------------------------------------------------------------- Hamming
def test_hamming_known_answers():
    z = np.zeros(32, np.uint8)
    f = np.full(32, 255, np.uint8)
    assert orc.hamming(z, z) == 0
    assert orc.hamming(z, f) == 256
    for bit in range(256):
        a = z.copy()
        a[bit // 8] = 1 << (bit % 8)
        assert orc.hamming(a, z) == 1
        assert orc.hamming(a, z, popcnt=True) == 1
    rng = np.random.default_rng(1)
    for _ in range(200):
        a = rng.integers(0, 256, 32, dtype=np.uint8)
        b = rng.integers(0, 256, 32, dtype=np.uint8)
        want = int(np.unpackbits(a ^ b).sum())
        assert orc.hamming(a, b) == want == orc.hamming(a, b, popcnt=True)


def test_knn2_tie_rules():
    # strict '<': lowest index wins a tie for best; a tie with best sets second = best
    q = np.zeros((1, 32), np.uint8)
    db = np.zeros((4, 32), np.uint8)
    db[0, 0] = 0b111      # 3
    db[1, 0] = 0b1        # 1
    db[2, 0] = 0b10       # 1 (tie, later)
    db[3, 0] = 0b11       # 2
    idx, d1, d2 = orc.knn2(q, db)
    assert (idx[0], d1[0], d2[0]) == (1, 1, 1)
    idx, d1, d2 = orc.knn2(q, db[:1])
    assert (idx[0], d1[0], d2[0]) == (0, 3, 256)
    idx, d1, d2 = orc.knn2(q, db[:0])
    assert (idx[0], d1[0], d2[0]) == (-1, 256, 256)
    assert orc.lib().orc_ratio_accept(50, 100, 50, 1, 0.6) == 1
    assert orc.lib().orc_ratio_accept(50, 100, 50, 0, 0.6) == 0
    # float compare: 0.6f*50.f rounds to 30.000002, so 30 passes (it would fail in exact arithmetic)
    assert orc.lib().orc_ratio_accept(30, 50, 50, 1, 0.6) == 1
    assert orc.lib().orc_ratio_accept(31, 50, 50, 1, 0.6) == 0


def test_ratio_float_compare():
    # (float)30 < 0.6f*50.f : 0.6f = 0.60000002384, product rounds to 30.000002 -> accepted
    assert np.float32(30) < np.float32(0.6) * np.float32(50)


# ---------------------------------------------------------------- rounding / atan
def test_cv_round_half_even():
    for v, want in [(0.5, 0), (1.5, 2), (2.5, 2), (-0.5, 0), (-1.5, -2), (2.4999, 2), (2.5001, 3)]:
        assert orc.lib().orc_cv_round_f(v) == want


def test_fast_atan2_known_answers():
    f = orc.lib().orc_fast_atan2
    assert f(0, 1) == 0
    assert f(1, 0) == 90
    assert f(0, -1) == 180
    assert f(-1, 0) == 270
    assert f(0, 0) == 0
    k = np.float32(180.0 / np.pi)
    p = [np.float32(c) * k for c in (0.9997878412794807, -0.3258083974640975, 0.1555786518463281,
                                     -0.04432655554792128)]
    c = np.float32(1) / (np.float32(1) + np.float32(2.220446049250313e-16))
    c2 = c * c
    want = (((p[3] * c2 + p[2]) * c2 + p[1]) * c2 + p[0]) * c
    assert f(1, 1) == want
    assert abs(f(1, 1) - 44.9904) < 1e-3           # the polynomial's 45 degrees
    rng = np.random.default_rng(2)
    for _ in range(2000):
        y, x = rng.normal(size=2).astype(np.float32) * 1000
        true = np.degrees(np.arctan2(float(y), float(x))) % 360
        got = f(y, x)
        assert min(abs(got - true), 360 - abs(got - true)) < 0.02


# ---------------------------------------------------------------- tables
def test_tables():
    e = orc.Extractor(1000, 1.2, 8, 20, 7)
    assert list(e.features_per_level()) == [217, 181, 151, 126, 105, 87, 73, 60]
    assert list(e.umax()) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert list(orc.Extractor(1200).features_per_level()) == [261, 217, 181, 151, 126, 105, 87, 72]
    assert list(orc.Extractor(2000).features_per_level()) == [434, 362, 302, 251, 209, 175, 145, 122]
    s, inv, sg, isg = e.scale_factors()
    assert s[0] == 1 and s[1] == np.float32(1.2) and s[2] == np.float32(np.float32(1.2) * np.float32(1.2))
    assert np.all(inv == np.float32(1) / s) and np.all(sg == s * s)


def test_level_sizes():
    e = orc.Extractor()
    e.extract(np.zeros((480, 752), np.uint8))
    assert [e.level_size(l) for l in range(8)] == [(752, 480), (627, 400), (522, 333), (435, 278),
                                                   (363, 231), (302, 193), (252, 161), (210, 134)]
    e = orc.Extractor(2000)
    e.extract(np.zeros((1080, 1920), np.uint8))
    assert [e.level_size(l) for l in range(8)] == [(1920, 1080), (1600, 900), (1333, 750), (1111, 625),
                                                   (926, 521), (772, 434), (643, 362), (536, 301)]


# ---------------------------------------------------------------- resize
def resize_numpy(src, dw, dh):
    """Independent numpy restatement of the 11-bit fixed-point bilinear resize."""
    sh, sw = src.shape

    def tab(ssize, dsize, clamp):
        scale = 1.0 / (np.float64(dsize) / ssize)
        d = np.arange(dsize)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = f - s.astype(np.float32)
        if clamp:
            f = np.where(s < 0, np.float32(0), f)
            s = np.maximum(s, 0)
            hi = s >= ssize - 1
            f = np.where(hi, np.float32(0), f)
            s = np.where(hi, ssize - 1, s)
        w0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        w1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return s, w0, w1

    sx, a0, a1 = tab(sw, dw, True)
    sy, b0, b1 = tab(sh, dh, False)
    S = src.astype(np.int64)
    sx1 = np.minimum(sx + 1, sw - 1)
    H = S[:, sx] * a0 + S[:, sx1] * a1                     # (sh, dw)
    r0 = H[np.clip(sy, 0, sh - 1)]
    r1 = H[np.clip(sy + 1, 0, sh - 1)]
    v = (((b0[:, None] * (r0 >> 4)) >> 16) + ((b1[:, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def test_resize_constant_and_numpy():
    for v in (0, 1, 77, 128, 254, 255):
        out = orc.resize(np.full((480, 752), v, np.uint8), 627, 400)
        assert np.all(out == v)
    rng = np.random.default_rng(3)
    for (sw, sh, dw, dh) in [(752, 480, 627, 400), (627, 400, 522, 333), (101, 67, 84, 56), (40, 30, 33, 25)]:
        src = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
        assert np.array_equal(orc.resize(src, dw, dh), resize_numpy(src, dw, dh))


# ---------------------------------------------------------------- blur
def blur_numpy(src, taps):
    K = np.array(taps, np.int64)
    p = np.pad(src.astype(np.int64), 3, mode="reflect")      # numpy 'reflect' == BORDER_REFLECT_101
    h, w = src.shape
    t = sum(K[k] * p[:, k:k + w] for k in range(7))
    t = np.minimum(t, 65535)
    v = sum(K[k] * t[k:k + h, :] for k in range(7))
    return np.minimum((v + 32768) >> 16, 255).astype(np.uint8)


def test_blur_known_answers():
    for v in (0, 1, 128, 200, 255):
        img = np.full((40, 50), v, np.uint8)
        assert np.all(orc.gaussian7(img, orc.BLUR_CV_4_2_PLUS) == v)
        assert np.all(orc.gaussian7(img, orc.BLUR_CV_4_0_0) == min(255, (66049 * v + 32768) >> 16))
    assert min(255, (66049 * 128 + 32768) >> 16) == 129
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    assert np.array_equal(orc.gaussian7(img, orc.BLUR_CV_4_0_0), blur_numpy(img, [18, 34, 49, 55, 49, 34, 18]))
    assert np.array_equal(orc.gaussian7(img, orc.BLUR_CV_4_2_PLUS), blur_numpy(img, [18, 34, 48, 56, 48, 34, 18]))


# ---------------------------------------------------------------- FAST
CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2),
          (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def strength_map(img):
    """S(p) = max over the 16 arcs of 9 of min(d) and of min(-d), d = centre - circle pixel; defined on the
    interior (3-px margin), 0 elsewhere.  p is a FAST corner at threshold t iff S > t; its score is S-1."""
    h, w = img.shape
    I = img.astype(np.int32)
    c = I[3:h - 3, 3:w - 3]
    d = np.stack([c - I[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in CIRCLE])   # (16, h-6, w-6)
    d2 = np.concatenate([d, d[:8]])
    A = np.max(np.stack([d2[k:k + 9].min(0) for k in range(16)]), 0)
    B = np.max(np.stack([(-d2[k:k + 9]).min(0) for k in range(16)]), 0)
    S = np.zeros((h, w), np.int32)
    S[3:h - 3, 3:w - 3] = np.maximum(A, B)
    return S


def fast_numpy(img, t):
    S = strength_map(img)
    sc = np.where(S > t, S - 1, 0)
    p = np.pad(sc, 1)
    h, w = img.shape
    nb = np.max(np.stack([p[1 + dy:1 + dy + h, 1 + dx:1 + dx + w] for dy in (-1, 0, 1) for dx in (-1, 0, 1)
                          if (dx, dy) != (0, 0)]), 0)
    ys, xs = np.nonzero((S > t) & (sc > nb))
    return [(int(x), int(y), int(sc[y, x])) for y, x in zip(ys, xs)]


def test_fast_single_pixel():
    img = np.zeros((21, 21), np.uint8)
    img[10, 10] = 255
    for t in (1, 7, 20, 100, 254):
        k = orc.fast(img, t)
        assert [(p["x"], p["y"], p["response"]) for p in k] == [(10, 10, 254)]
        assert k[0]["size"] == 7 and k[0]["angle"] == -1 and k[0]["octave"] == 0 and k[0]["class_id"] == -1
    img[10, 11] = 255      # equal neighbours suppress each other
    assert len(orc.fast(img, 20)) == 0
    assert len(orc.fast(img, 7)) == 0
    assert len(orc.fast(img, 20, nms=False)) == 2


def test_fast_matches_strength_definition():
    rng = np.random.default_rng(5)
    for trial in range(6):
        h, w = rng.integers(20, 60, 2)
        base = rng.integers(0, 256, (h // 4 + 2, w // 4 + 2)).repeat(4, 0).repeat(4, 1)[:h, :w]
        img = np.clip(base + rng.integers(-12, 13, (h, w)), 0, 255).astype(np.uint8)
        for t in (7, 20, 40):
            got = orc.fast(img, t)
            want = fast_numpy(img, t)
            assert [(int(p["x"]), int(p["y"]), int(p["response"])) for p in got] == want
            S = strength_map(img)
            nn = orc.fast(img, t, nms=False)
            ys, xs = np.nonzero(S > t)
            assert [(int(p["x"]), int(p["y"])) for p in nn] == [(int(x), int(y)) for y, x in zip(ys, xs)]


# ---------------------------------------------------------------- quadtree
def test_octree_properties():
    rng = np.random.default_rng(6)
    W, H = 720, 448
    for n, N in [(0, 50), (1, 50), (5, 50), (300, 50), (3000, 217), (3000, 5000), (217, 217)]:
        pts = set()
        while len(pts) < n:
            pts.add((int(rng.integers(3, W - 3)), int(rng.integers(3, H - 3))))
        kps = np.zeros(n, orc.KP)
        for i, (x, y) in enumerate(sorted(pts, key=lambda p: (p[1] // 32, p[0] // 30, p[1], p[0]))):
            kps[i] = (x, y, 7, -1, int(rng.integers(7, 255)), 0, -1)
        out = orc.distribute_octree(kps, 16, 16 + W, 16, 16 + H, N)
        assert len(out) <= min(n, N + 3)
        if n <= 1 or N >= 4 * n:
            assert len(out) == n                      # every point ends alone in a node
        elif n >= N:
            assert len(out) >= min(N, n) or len(out) == n
        src = {(p["x"], p["y"]): p["response"] for p in kps}
        assert all(src[(p["x"], p["y"])] == p["response"] for p in out)
        assert len({(p["x"], p["y"]) for p in out}) == len(out)
        again = orc.distribute_octree(kps, 16, 16 + W, 16, 16 + H, N)
        assert np.array_equal(out, again)


def test_octree_keeps_strongest_first_on_ties():
    # two points in one final node with equal response: the earlier one is kept
    kps = np.zeros(3, orc.KP)
    kps[0] = (10, 10, 7, -1, 50, 0, -1)
    kps[1] = (11, 11, 7, -1, 50, 0, -1)
    kps[2] = (500, 300, 7, -1, 20, 0, -1)
    out = orc.distribute_octree(kps, 16, 736, 16, 464, 2)
    assert len(out) == 2
    assert (out["x"].tolist().count(10), out["x"].tolist().count(11)) == (1, 0)


# ---------------------------------------------------------------- whole extractor
def test_extract_degenerate_and_basic():
    import my_orb_slam2_amd as m
    e = orc.Extractor()
    for kind in (m.synth.ZERO, m.synth.CONST128):
        k, d = e.extract(m.synth_frame(0, kind=kind))
        assert len(k) == 0
    k, d = e.extract(None)
    assert len(k) == 0
    img = m.synth_frame(0)
    k, d = e.extract(img)
    assert 900 <= len(k) <= 1000 + 3 * 8
    assert np.all(np.diff(k["octave"]) >= 0)                 # levels are concatenated in order
    assert np.all((k["angle"] >= 0) & (k["angle"] < 360))
    assert np.all(k["class_id"] == -1)
    s = e.scale_factors()[0]
    assert np.all(k["size"] == np.floor(31 * s[k["octave"]]))
    # keypoints stay 19 px inside their level
    for l in range(8):
        w, h = e.level_size(l)
        sel = e.level_selected(l)
        assert np.all((sel["x"] >= 19) & (sel["x"] <= w - 20) & (sel["y"] >= 19) & (sel["y"] <= h - 20))
    # the padded level is the level with a BORDER_REFLECT_101 frame
    for l in (0, 3):
        assert np.array_equal(e.level_image(l, padded=True), np.pad(e.level_image(l), 19, mode="reflect"))
    # descriptor of a keypoint re-derived from the blurred level
    k2, d2 = e.extract(img)
    assert np.array_equal(k, k2) and np.array_equal(d, d2)


# ---------------------------------------------------------------- stored regression vectors
def test_oracle_reproduces_golden_vectors():
    import os
    import my_orb_slam2_amd as m
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name, idx, w, h, nf in [("extract_752x480_f0", 0, 752, 480, 1000), ("extract_321x243_f11", 11, 321, 243, 500)]:
        g = np.load(os.path.join(gdir, name + ".npz"))
        e = orc.Extractor(nf, 1.2, 8, 20, 7, orc.BLUR_CV_4_0_0)
        k, d = e.extract(m.synth_frame(idx, w, h))
        assert np.array_equal(k, g["keypoints"]) and np.array_equal(d, g["descriptors"])
        assert [list(e.level_size(l)) for l in range(8)] == g["level_sizes"].tolist()
    g = np.load(os.path.join(gdir, "knn2_256x4096.npz"))
    idx, d1, d2 = orc.knn2(g["query"], g["db"])
    assert np.array_equal(idx, g["idx"]) and np.array_equal(d1, g["d1"]) and np.array_equal(d2, g["d2"])


# ---------------------------------------------------------------- undistort + grid
def test_undistort_known_answers():
    # no distortion coefficients: undistortPoints is the identity up to float rounding of (u-cx)/fx*fx+cx
    K4 = [458.654, 457.296, 367.215, 248.375]
    pts = np.array([[367.215, 248.375], [0, 0], [751, 479]], np.float32)
    out = orc.undistort_points(pts, K4, [0, 0, 0, 0, 0])
    assert np.abs(out - pts).max() < 1e-4
    # the principal point is a fixed point of the distortion model
    out = orc.undistort_points(pts[:1], K4, [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0])
    assert np.abs(out - pts[:1]).max() < 1e-4
    # barrel distortion (k1 < 0): undistorted corners move outwards
    out = orc.undistort_points(pts[1:], K4, [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0])
    assert out[0, 0] < 0 and out[0, 1] < 0 and out[1, 0] > 751 and out[1, 1] > 479


def test_assign_grid_known_answers():
    kps = np.zeros(5, orc.KP)
    kps["x"] = [0, 751.9, 376, 376, -30]
    kps["y"] = [0, 479.9, 240, 240, 10]
    start, items = orc.assign_grid(kps, (0.0, 0.0, 752.0, 480.0))
    # round(x * 64/752): 0 -> 0; 751.9 -> 64 (outside, dropped); 376 -> 32; -30 -> -3 (dropped)
    assert start[-1] == 3 and list(items) == [0, 2, 3]
    assert start[32 * 48 + 24 + 1] - start[32 * 48 + 24] == 2


def test_search_by_bow_against_python_restatement():
    """orc_search_by_bow against a direct Python transcription of src/ORBmatcher.cc:153-293 / :532-690 on a
    small case (dict of node -> list, bisect for lower_bound)."""
    import bisect
    from tests.test_gpu_search_bow import make_case
    k1, d1, v1, fv1, k2, d2, v2, fv2 = make_case(11, 150, 160, 12, noise_bits=30)
    ham = lambda a, b: int(np.unpackbits(a ^ b).sum())

    def restate(mode, ratio, check_ori):
        keys1, keys2 = list(fv1[0]), list(fv2[0])
        lst1 = {int(n): list(fv1[2][fv1[1][j]:fv1[1][j + 1]]) for j, n in enumerate(keys1)}
        lst2 = {int(n): list(fv2[2][fv2[1][j]:fv2[1][j + 1]]) for j, n in enumerate(keys2)}
        out = [-1] * (len(k2) if mode == 0 else len(k1))
        matched2 = [False] * len(k2)
        hist = [[] for _ in range(30)]
        factor = np.float32(1.0) / np.float32(30)
        a = b = 0
        n = 0
        while a < len(keys1) and b < len(keys2):
            if keys1[a] == keys2[b]:
                for idx1 in lst1[int(keys1[a])]:
                    if not v1[idx1]:
                        continue
                    best1, best2, bidx = 256, 256, -1
                    for idx2 in lst2[int(keys2[b])]:
                        if mode == 0 and out[idx2] >= 0:
                            continue
                        if mode == 1 and (matched2[idx2] or not v2[idx2]):
                            continue
                        dist = ham(d1[idx1], d2[idx2])
                        if dist < best1:
                            best2, best1, bidx = best1, dist, int(idx2)
                        elif dist < best2:
                            best2 = dist
                    ok = best1 <= 50 if mode == 0 else best1 < 50
                    if ok and np.float32(best1) < np.float32(ratio) * np.float32(best2):
                        if mode == 0:
                            out[bidx] = int(idx1)
                        else:
                            out[idx1] = bidx
                            matched2[bidx] = True
                        if check_ori:
                            rot = np.float32(k1["angle"][idx1]) - np.float32(k2["angle"][bidx])
                            if rot < 0:
                                rot = np.float32(rot + np.float32(360.0))
                            q = float(np.float32(rot / (np.float32(360) * factor)))
                            bin_ = int(np.floor(q + 0.5)) if q >= 0 else int(np.ceil(q - 0.5))
                            if bin_ == 30:
                                bin_ = 0
                            hist[bin_].append(bidx if mode == 0 else int(idx1))
                        n += 1
                a += 1
                b += 1
            elif keys1[a] < keys2[b]:
                a = bisect.bisect_left(keys1, keys2[b])
            else:
                b = bisect.bisect_left(keys2, keys1[a])
        if check_ori:
            sizes = [len(h) for h in hist]
            m1 = m2 = m3 = 0
            i1 = i2 = i3 = -1
            for i, s in enumerate(sizes):
                if s > m1:
                    m3, m2, m1 = m2, m1, s
                    i3, i2, i1 = i2, i1, i
                elif s > m2:
                    m3, m2 = m2, s
                    i3, i2 = i2, i
                elif s > m3:
                    m3, i3 = s, i
            if m2 < np.float32(0.1) * np.float32(m1):
                i2 = i3 = -1
            elif m3 < np.float32(0.1) * np.float32(m1):
                i3 = -1
            for i in range(30):
                if i in (i1, i2, i3):
                    continue
                for e in hist[i]:
                    out[e] = -1
                    n -= 1
        return n, np.array(out, np.int32)

    seen = 0
    for mode in (0, 1):
        for ratio in (0.7, 0.9):
            for check_ori in (False, True):
                want_n, want = restate(mode, ratio, check_ori)
                got_n, got = orc.search_by_bow(k1, d1, v1, fv1, k2, d2, fv2, v2 if mode else None, ratio, check_ori)
                assert got_n == want_n and np.array_equal(got, want)
                seen += want_n
    assert seen > 20


def test_search_by_projection_against_python_restatement():
    """orc_search_by_projection_frame against a direct Python transcription of src/ORBmatcher.cc:1395-1513 with
    Frame::GetFeaturesInArea (src/Frame.cc:365-418) on random keypoints (float32 arithmetic spelled out)."""
    f32 = np.float32
    rng = np.random.default_rng(3)
    n2, npts = 400, 260
    bounds = (f32(0), f32(0), f32(752), f32(480))
    k2 = np.zeros(n2, orc.KP)
    k2["x"] = rng.uniform(0, 752, n2).astype(f32)
    k2["y"] = rng.uniform(0, 480, n2).astype(f32)
    k2["octave"] = rng.integers(0, 8, n2)
    k2["angle"] = rng.uniform(0, 360, n2).astype(f32)
    d2 = rng.integers(0, 256, (n2, 32), dtype=np.uint8)
    src = rng.integers(0, n2, npts)
    uv = np.stack([k2["x"][src], k2["y"][src]], 1).astype(f32) + rng.normal(0, 3, (npts, 2)).astype(f32)
    uv[rng.random(npts) < 0.05] += f32(800)
    inv_z = rng.uniform(0.05, 1, npts).astype(f32)
    inv_z[rng.random(npts) < 0.05] *= -1
    octs = np.clip(k2["octave"][src] + rng.integers(-1, 2, npts), 0, 7).astype(np.int32)
    ang = np.mod(k2["angle"][src] + np.where(rng.random(npts) < 0.7, 20, rng.uniform(0, 360, npts)), 360).astype(f32)
    dm = d2[src] ^ (rng.integers(0, 256, (npts, 32), dtype=np.uint8) & rng.integers(0, 256, (npts, 32), dtype=np.uint8)
                    & 0x11)
    has_obs = (rng.random(npts) < 0.8).astype(np.uint8)
    occupied = (rng.random(n2) < 0.1).astype(np.uint8)
    ur2 = np.where(rng.random(n2) < 0.5, k2["x"] - rng.uniform(0, 30, n2), -1).astype(f32)
    sf = np.cumprod([1.0] + [1.2] * 7).astype(f32)
    ham = lambda a, b: int(np.unpackbits(a ^ b).sum())
    minX, minY, maxX, maxY = bounds
    invW, invH = f32(64) / f32(maxX - minX), f32(48) / f32(maxY - minY)
    rnd = lambda q: int(np.floor(q + 0.5)) if q >= 0 else int(np.ceil(q - 0.5))
    grid = {}
    for i in range(n2):
        px, py = rnd(float(f32(f32(k2["x"][i] - minX) * invW))), rnd(float(f32(f32(k2["y"][i] - minY) * invH)))
        if 0 <= px < 64 and 0 <= py < 48:
            grid.setdefault((px, py), []).append(i)

    def area(x, y, r, lo, hi):
        out = []
        x0 = max(0, int(np.floor(f32(f32(f32(x - minX) - r) * invW))))
        x1 = min(63, int(np.ceil(f32(f32(f32(x - minX) + r) * invW))))
        y0 = max(0, int(np.floor(f32(f32(f32(y - minY) - r) * invH))))
        y1 = min(47, int(np.ceil(f32(f32(f32(y - minY) + r) * invH))))
        if x0 >= 64 or x1 < 0 or y0 >= 48 or y1 < 0:
            return out
        check = lo > 0 or hi >= 0
        for ix in range(x0, x1 + 1):
            for iy in range(y0, y1 + 1):
                for j in grid.get((ix, iy), []):
                    o = int(k2["octave"][j])
                    if check and (o < lo or (hi >= 0 and o > hi)):
                        continue
                    if abs(f32(k2["x"][j] - x)) < r and abs(f32(k2["y"][j] - y)) < r:
                        out.append(j)
        return out

    def restate(th, mbf, direction, stereo, check_ori, max_dist=100):
        holder = [-1] * n2
        blocked = [bool(b) for b in occupied]
        hist = [[] for _ in range(30)]
        factor = f32(1.0) / f32(30)
        n = 0
        for i in range(npts):
            iz, u, v = inv_z[i], uv[i, 0], uv[i, 1]
            if iz < 0 or u < minX or u > maxX or v < minY or v > maxY:
                continue
            o = int(octs[i])
            radius = f32(f32(th) * sf[o])
            lo, hi = ((o, -1) if direction == 1 else (0, o) if direction == 2 else (o - 1, o + 1))
            best, bidx = 256, -1
            for j in area(u, v, radius, lo, hi):
                if blocked[j]:
                    continue
                if stereo and ur2[j] > 0:
                    ur = f32(u - f32(f32(mbf) * iz))
                    if abs(f32(ur - ur2[j])) > radius:
                        continue
                dist = ham(dm[i], d2[j])
                if dist < best:
                    best, bidx = dist, j
            if best <= max_dist:
                holder[bidx] = i
                blocked[bidx] = bool(has_obs[i])
                n += 1
                if check_ori:
                    rot = f32(ang[i] - k2["angle"][bidx])
                    if rot < 0:
                        rot = f32(rot + f32(360))
                    b = rnd(float(f32(rot / f32(f32(360) * factor))))
                    hist[0 if b == 30 else b].append(bidx)
        if check_ori:
            m1 = m2 = m3 = 0
            i1 = i2 = i3 = -1
            for i, h in enumerate(hist):
                s = len(h)
                if s > m1:
                    m3, m2, m1, i3, i2, i1 = m2, m1, s, i2, i1, i
                elif s > m2:
                    m3, m2, i3, i2 = m2, s, i2, i
                elif s > m3:
                    m3, i3 = s, i
            if m2 < f32(0.1) * f32(m1):
                i2 = i3 = -1
            elif m3 < f32(0.1) * f32(m1):
                i3 = -1
            for i in range(30):
                if i not in (i1, i2, i3):
                    for e in hist[i]:
                        holder[e] = -1
                        n -= 1
        return n, np.array(holder, np.int32)

    # --- the map-point overload (src/ORBmatcher.cc:41-123) on the same data ---
    view_cos = rng.uniform(0.99, 1.0, npts).astype(f32)
    xr = (uv[:, 0] - rng.uniform(0, 30, npts)).astype(f32)

    def restate_points(th, ratio, stereo):
        holder = [-1] * n2
        blocked = [bool(b) for b in occupied]
        n = 0
        for i in range(npts):
            o = int(octs[i])
            r = f32(2.5) if float(view_cos[i]) > 0.998 else f32(4.0)
            if th != 1.0:
                r = f32(r * f32(th))
            radius = f32(r * sf[o])
            b1, l1, b2, l2, bi = 256, -1, 256, -1, -1
            for j in area(uv[i, 0], uv[i, 1], radius, o - 1, o):
                if blocked[j]:
                    continue
                if stereo and ur2[j] > 0 and abs(f32(xr[i] - ur2[j])) > radius:
                    continue
                dist = ham(dm[i], d2[j])
                if dist < b1:
                    b2, b1, l2, l1, bi = b1, dist, l1, int(k2["octave"][j]), j
                elif dist < b2:
                    l2, b2 = int(k2["octave"][j]), dist
            if b1 <= 100:
                if l1 == l2 and f32(b1) > f32(f32(ratio) * f32(b2)):
                    continue
                holder[bi] = i
                blocked[bi] = bool(has_obs[i])
                n += 1
        return n, np.array(holder, np.int32)

    seen_pts = 0
    for th, ratio, stereo in ((1.0, 0.8, False), (3.0, 0.8, True), (5.0, 0.6, True)):
        want_n, want = restate_points(th, ratio, stereo)
        got_n, got = orc.search_by_projection_points(uv, view_cos, octs, dm, has_obs, k2, d2, bounds, sf, th, ratio,
                                                     xr if stereo else None, ur2 if stereo else None, occupied)
        assert got_n == want_n and np.array_equal(got, want), (th, ratio, stereo)
        seen_pts += want_n
    assert seen_pts > 60

    seen = 0
    for th, mbf, direction, stereo, check_ori, md in ((15, 0.0, 0, False, True, 100), (7, 40.0, 0, True, True, 100),
                                                      (14, 40.0, 1, True, False, 100), (14, 40.0, 2, True, True, 100),
                                                      (40, 0.0, 0, False, False, 100), (10, 0.0, 0, False, True, 64)):
        want_n, want = restate(th, mbf, direction, stereo, check_ori, md)
        got_n, got = orc.search_by_projection_frame(uv, inv_z, octs, ang, dm, has_obs, k2, d2, bounds, sf, th, mbf,
                                                    direction, ur2 if stereo else None, occupied, check_ori, md)
        assert got_n == want_n and np.array_equal(got, want), (th, direction, stereo, check_ori)
        seen += max(want_n, 0)
    assert seen > 100
