"""GPU parity: Frame::UndistortKeyPoints + AssignFeaturesToGrid against the oracle (bit-exact floats:
both sides evaluate cv::undistortPoints' double-precision iteration without fused multiply-adds)."""
import numpy as np
import pytest

import my_orb_slam2_amd as m
from tests import oracle as orc

pytestmark = pytest.mark.gpu

# EuRoC mono (example_data/EuRoC_mono.yaml): Camera.fx fy cx cy k1 k2 p1 p2, k3 = 0
EUROC = dict(fx=458.654, fy=457.296, cx=367.215, cy=248.375, k1=-0.28340811, k2=0.07395907, p1=0.00019359,
             p2=1.76187114e-05, k3=0.0)


def run(cam_kw, frames_idx):
    import torch
    cam = m.Camera(**cam_kw)
    K4 = [cam_kw["fx"], cam_kw["fy"], cam_kw["cx"], cam_kw["cy"]]
    D5 = [cam_kw["k1"], cam_kw["k2"], cam_kw["p1"], cam_kw["p2"], cam_kw["k3"]]
    n = len(frames_idx)
    ex = m.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=n)
    cap = ex.capacity
    frames = np.stack([m.synth_frame(i) for i in frames_idx])
    frames[-1] = 0                                    # a frame without keypoints
    d_kps = torch.zeros((n, cap, 7), dtype=torch.float32, device="cuda")
    d_desc = torch.zeros((n, cap, 32), dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
    ex.extract_batch_device(torch.from_numpy(frames).cuda(), d_kps, d_desc, d_cnt)
    ex.synchronize()
    hm = m.HammingMatcher()
    bounds = hm.image_bounds(cam, 752, 480)
    if cam_kw["k1"] != 0:
        c = orc.undistort_points([[0, 0], [752, 0], [0, 480], [752, 480]], K4, D5)
        want = (min(c[0, 0], c[2, 0]), min(c[0, 1], c[1, 1]), max(c[1, 0], c[3, 0]), max(c[2, 1], c[3, 1]))
        assert bounds == tuple(float(v) for v in want)
    else:
        assert bounds == (0.0, 0.0, 752.0, 480.0)
    d_un = torch.zeros_like(d_kps)
    d_start = torch.zeros((n, 64 * 48 + 1), dtype=torch.int32, device="cuda")
    d_items = torch.full((n, cap), -1, dtype=torch.int32, device="cuda")
    hm.undistort_assign_grid_device(d_kps, d_cnt, cam, bounds, d_un, d_start, d_items)
    hm.synchronize()
    cnt = d_cnt.cpu().numpy()
    view = lambda t: t.cpu().numpy().view(np.uint8).reshape(n, cap, 28).copy().view(m.KEYPOINT_DTYPE).reshape(n, cap)
    kps, un = view(d_kps), view(d_un)
    start, items = d_start.cpu().numpy(), d_items.cpu().numpy()
    for f in range(n):
        k = kps[f, :cnt[f]]
        want = k.copy()
        if cam_kw["k1"] != 0 and len(k):
            xy = orc.undistort_points(np.stack([k["x"], k["y"]], 1), K4, D5)
            want["x"], want["y"] = xy[:, 0], xy[:, 1]
        assert np.array_equal(un[f, :cnt[f]], want), "mvKeysUn differs in frame %d" % f
        ostart, oitems = orc.assign_grid(want, bounds)
        assert np.array_equal(start[f], ostart)
        assert np.array_equal(items[f, :ostart[-1]], oitems)
    assert cnt[-1] == 0 and start[-1, -1] == 0
    return cnt


def test_undistort_and_grid_euroc_mono():
    cnt = run(EUROC, [70, 71, 72, 73])
    assert cnt[0] > 900


def test_grid_without_distortion():
    run(dict(EUROC, k1=0.0, k2=0.0, p1=0.0, p2=0.0), [74, 75])


def test_search_for_initialization():
    """Mono initialisation matching between two views of the same scene (second view: objects shifted by
    their disparity), through undistort + grid + SearchForInitialization, against the oracle."""
    cam = dict(EUROC, k1=0.0, k2=0.0, p1=0.0, p2=0.0)
    bounds = (0.0, 0.0, 752.0, 480.0)
    ex = m.ORBextractor(2000, 1.2, 8, 20, 7)           # the initialisation extractor uses 2 * nFeatures
    hm = m.HammingMatcher()
    for idx, check_ori in ((80, True), (81, False)):
        k1, d1 = ex(m.synth_frame(idx))
        k2, d2 = ex(m.synth_frame(idx, right_view=True))
        start, items = orc.assign_grid(k2, bounds)
        prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32).copy()   # mvbPrevMatched = keypoint positions
        want_n, want, want_prev = orc.search_for_initialization(k1, d1, k2, d2, bounds, prev, 100, 0.9, check_ori)
        n, got = hm.search_for_initialization(k1, d1, k2, d2, start, items, bounds, prev, 100, 0.9, check_ori)
        assert n == want_n and np.array_equal(got, want)
        assert np.array_equal(prev, want_prev)
        assert n > 20 and (got >= 0).sum() == n
    # empty second frame
    n, got = hm.search_for_initialization(k1, d1, k2[:0], d2[:0], np.zeros(64 * 48 + 1, np.int32), np.zeros(1, np.int32),
                                          bounds, prev)
    assert n == 0 and np.all(got == -1)


def test_search_for_initialization_refuses_a_malformed_grid():
    """Indices the kernel would follow are checked on the host: out of range, repeated, starts that fall."""
    bounds = (0.0, 0.0, 752.0, 480.0)
    ex = m.ORBextractor(500, 1.2, 8, 20, 7)
    hm = m.HammingMatcher()
    k1, d1 = ex(m.synth_frame(82))
    k2, d2 = ex(m.synth_frame(82, right_view=True))
    start, items = orc.assign_grid(k2, bounds)
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32).copy()
    n_ok, _ = hm.search_for_initialization(k1, d1, k2, d2, start, items, bounds, prev.copy())
    assert n_ok > 0
    bad_items = items.copy()
    bad_items[3] = len(k2) + 7
    bad_dup = items.copy()
    bad_dup[5] = bad_dup[4]
    bad_start = start.copy()
    bad_start[100] = bad_start[99] - 1 if bad_start[99] > 0 else -1
    for s, it in ((start, bad_items), (start, bad_dup), (bad_start, items)):
        with pytest.raises(m.OrbxError):
            hm.search_for_initialization(k1, d1, k2, d2, s, it, bounds, prev.copy())


def test_search_for_initialization_on_device_resident_frames():
    """The whole chain without the frames' data leaving the GPU: batched extraction of both views, undistort +
    grid on the device, then orbx_search_for_initialization_device on the descriptor / keypoint / grid slices of
    the two frames; only the host copies of mvKeysUn (which a Frame holds anyway) go in from the host side.
    Compared with the oracle and with the host-array form."""
    import torch
    cam = m.Camera(**dict(EUROC, k1=0.0, k2=0.0, p1=0.0, p2=0.0))
    bounds = (0.0, 0.0, 752.0, 480.0)
    ex = m.ORBextractor(2000, 1.2, 8, 20, 7, max_batch=2)
    hm = m.HammingMatcher()
    cap = ex.capacity
    for idx, check_ori, window in ((83, True, 100), (84, False, 100), (85, True, 400)):
        frames = np.stack([m.synth_frame(idx), m.synth_frame(idx, right_view=True)])
        d_kps = torch.zeros((2, cap, 7), dtype=torch.float32, device="cuda")
        d_desc = torch.zeros((2, cap, 32), dtype=torch.uint8, device="cuda")
        d_cnt = torch.zeros(2, dtype=torch.int32, device="cuda")
        ex.extract_batch_device(torch.from_numpy(frames).cuda(), d_kps, d_desc, d_cnt)
        ex.synchronize()
        d_un = torch.zeros_like(d_kps)
        d_start = torch.zeros((2, 64 * 48 + 1), dtype=torch.int32, device="cuda")
        d_items = torch.full((2, cap), -1, dtype=torch.int32, device="cuda")
        hm.undistort_assign_grid_device(d_kps, d_cnt, cam, bounds, d_un, d_start, d_items)
        hm.synchronize()
        cnt = d_cnt.cpu().numpy()
        un = d_un.cpu().numpy().view(np.uint8).reshape(2, cap, 28).copy().view(m.KEYPOINT_DTYPE).reshape(2, cap)
        k1, k2 = un[0, :cnt[0]].copy(), un[1, :cnt[1]].copy()
        desc = d_desc.cpu().numpy()
        d1, d2 = desc[0, :cnt[0]].copy(), desc[1, :cnt[1]].copy()
        prev0 = np.stack([k1["x"], k1["y"]], 1).astype(np.float32).copy()
        want_prev = prev0.copy()
        want_n, want, want_prev = orc.search_for_initialization(k1, d1, k2, d2, bounds, want_prev, window, 0.9, check_ori)
        prev = prev0.copy()
        n, got = hm.search_for_initialization_device(k1, d_desc[0], k2, d_un[1], d_desc[1], d_start[1], d_items[1],
                                                     bounds, prev, window, 0.9, check_ori)
        assert n == want_n and np.array_equal(got, want) and np.array_equal(prev, want_prev)
        assert n > 20
        start, items = d_start[1].cpu().numpy(), d_items[1].cpu().numpy()[:cnt[1]]
        prev = prev0.copy()
        n2_, got2 = hm.search_for_initialization(k1, d1, k2, d2, start, items, bounds, prev, window, 0.9, check_ori)
        assert n2_ == want_n and np.array_equal(got2, want) and np.array_equal(prev, want_prev)


def test_whole_image_windows():
    """A window that covers the whole image makes every level-0 keypoint of frame 2 a candidate of every level-0
    query: millions of candidates in lists of thousands, far more than a wave holds at once."""
    bounds = (0.0, 0.0, 752.0, 480.0)
    ex = m.ORBextractor(5000, 1.2, 8, 20, 7)
    hm = m.HammingMatcher()
    k1, d1 = ex(m.synth_frame(86))
    k2, d2 = ex(m.synth_frame(86, right_view=True))
    n0 = int((k1["octave"] == 0).sum())
    assert n0 * int((k2["octave"] == 0).sum()) > 1 << 20
    start, items = orc.assign_grid(k2, bounds)
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32).copy()
    want_prev = prev.copy()
    want_n, want, want_prev = orc.search_for_initialization(k1, d1, k2, d2, bounds, want_prev, 2000, 0.9, True)
    n, got = hm.search_for_initialization(k1, d1, k2, d2, start, items, bounds, prev, 2000, 0.9, True)
    assert n == want_n and np.array_equal(got, want) and np.array_equal(prev, want_prev)


def test_device_resident_search_edges_and_a_corrupt_device_grid():
    """Empty frames give no matches; a device grid whose indices point outside frame 2 (something the host cannot
    check without copying it back) is walked without leaving the arrays: the bad entries are skipped."""
    import torch
    bounds = (0.0, 0.0, 752.0, 480.0)
    ex = m.ORBextractor(500, 1.2, 8, 20, 7)
    hm = m.HammingMatcher()
    k1, d1 = ex(m.synth_frame(87))
    k2, d2 = ex(m.synth_frame(87, right_view=True))
    start, items = orc.assign_grid(k2, bounds)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    t_d1, t_d2 = t(d1), t(d2)
    t_k2 = t(k2.view(np.uint8).reshape(-1, 28))
    prev0 = np.stack([k1["x"], k1["y"]], 1).astype(np.float32).copy()
    want_n, want, _ = orc.search_for_initialization(k1, d1, k2, d2, bounds, prev0.copy(), 100, 0.9, True)
    n, got = hm.search_for_initialization_device(k1, t_d1, k2, t_k2, t_d2, t(start), t(items.astype(np.int32)), bounds,
                                                 prev0.copy())
    assert n == want_n and np.array_equal(got, want)
    # no keypoints on one side or the other
    n, got = hm.search_for_initialization_device(k1, t_d1, k2[:0], t_k2, t_d2, t(start), t(items.astype(np.int32)), bounds,
                                                 prev0.copy())
    assert n == 0 and np.all(got == -1) and len(got) == len(k1)
    n, got = hm.search_for_initialization_device(k1[:0], t_d1, k2, t_k2, t_d2, t(start), t(items.astype(np.int32)), bounds,
                                                 prev0[:0].copy())
    assert n == 0 and len(got) == 0
    # corrupt grid: indices far outside frame 2, negative ones, and cell starts beyond the item array
    bad_items = items.astype(np.int32).copy()
    bad_items[::7] = 1 << 20
    bad_items[3::11] = -5
    bad_start = start.copy()
    bad_start[2000:] = 1 << 24
    n, got = hm.search_for_initialization_device(k1, t_d1, k2, t_k2, t_d2, t(bad_start), t(bad_items), bounds, prev0.copy())
    assert 0 <= n <= len(k1) and np.all((got >= -1) & (got < len(k2)))


@pytest.mark.parametrize("n_bases,n,window", [(6, 500, 400), (2, 300, 60), (30, 1500, 150)])
def test_search_for_initialization_crowded(n_bases, n, window):
    """Made-up frames where most descriptors are near copies of a few: nearly every candidate is within TH_LOW,
    ties in distance are everywhere, and the keypoints matched early keep later ones from their closest
    candidates. The ordered short lists the kernel returns rarely settle a query here, so the path that reads a
    query's whole list back is the one that decides; both call forms must still equal the oracle."""
    import torch
    rng = np.random.default_rng(500 + n_bases)
    bounds = (0.0, 0.0, 752.0, 480.0)
    bases = rng.integers(0, 256, (n_bases, 32), dtype=np.uint8)

    def frame(count):
        k = np.zeros(count, dtype=m.KEYPOINT_DTYPE)
        k["x"] = rng.uniform(200, 500, count).astype(np.float32)
        k["y"] = rng.uniform(100, 400, count).astype(np.float32)
        k["octave"] = (rng.random(count) < 0.15).astype(np.int32)        # a few above level 0
        k["angle"] = rng.uniform(0, 360, count).astype(np.float32)
        d = bases[rng.integers(0, n_bases, count)].copy()
        for i in range(count):
            for b in rng.choice(256, rng.integers(0, 12), replace=False):
                d[i, b >> 3] ^= 1 << (b & 7)
        return k, d

    k1, d1 = frame(n)
    k2, d2 = frame(n + 37)
    start, items = orc.assign_grid(k2, bounds)
    hm = m.HammingMatcher()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    t_d1, t_d2, t_k2 = t(d1), t(d2), t(k2.view(np.uint8).reshape(-1, 28))
    t_start, t_items = t(start), t(items.astype(np.int32))
    prev0 = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
    total = 0
    for ratio in (0.9, 0.6):
        for check_ori in (True, False):
            want_n, want, want_prev = orc.search_for_initialization(k1, d1, k2, d2, bounds, prev0.copy(), window, ratio,
                                                                    check_ori)
            prev = prev0.copy()
            n_got, got = hm.search_for_initialization(k1, d1, k2, d2, start, items, bounds, prev, window, ratio, check_ori)
            assert n_got == want_n and np.array_equal(got, want) and np.array_equal(prev, want_prev)
            prev = prev0.copy()
            n_got, got = hm.search_for_initialization_device(k1, t_d1, k2, t_k2, t_d2, t_start, t_items, bounds, prev,
                                                             window, ratio, check_ori)
            assert n_got == want_n and np.array_equal(got, want) and np.array_equal(prev, want_prev)
            total += want_n
    assert total > 0
