"""GPU parity of orbx_search_by_projection_frame against the oracle's restatement of
ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (src/ORBmatcher.cc:1372-1516)."""
import numpy as np
import pytest

import my_orb_slam2_amd as m
from tests import oracle as orc

pytestmark = pytest.mark.gpu

BOUNDS = (0.0, 0.0, 752.0, 480.0)


def make_case(seed, th, direction, stereo, frame=90, nfeatures=1000):
    """Last frame = a synthetic scene, current frame = the same scene seen from a little to the side (objects
    shifted by their disparity). Map points: a random 70 % of the last frame's features; their projections into
    the current frame are the last positions moved by a smooth flow plus noise, a few behind the camera or
    outside the image; descriptors are the last frame's with a few bits flipped. Some current features already
    hold a point; some map points have no observations (the temporal points of stereo tracking)."""
    rng = np.random.default_rng(seed)
    ex = m.ORBextractor(nfeatures, 1.2, 8, 20, 7)
    kl, dl = ex(m.synth_frame(frame))
    kc, dc = ex(m.synth_frame(frame, right_view=True))
    sf = ex.GetScaleFactors()
    pick = np.flatnonzero(rng.random(len(kl)) < 0.7)
    npts = len(pick)
    uv = np.stack([kl["x"][pick], kl["y"][pick]], 1).astype(np.float32)
    uv[:, 0] -= rng.uniform(0, 30, npts).astype(np.float32)              # disparity-like flow
    uv += rng.normal(0, 1.5, (npts, 2)).astype(np.float32)
    inv_z = rng.uniform(0.02, 1.0, npts).astype(np.float32)
    inv_z[rng.random(npts) < 0.03] *= -1                                   # behind the camera
    uv[rng.random(npts) < 0.03] += np.float32(900)                         # outside the image
    desc_mp = dl[pick].copy()
    for i in range(npts):
        for b in rng.choice(256, int(rng.integers(0, 25)), replace=False):
            desc_mp[i, b >> 3] ^= 1 << (b & 7)
    has_obs = (rng.random(npts) < 0.85).astype(np.uint8)
    occupied = (rng.random(len(kc)) < 0.1).astype(np.uint8)
    u_right = None
    mbf = 0.0
    if stereo:
        mbf = 47.9
        u_right = np.where(rng.random(len(kc)) < 0.6, kc["x"] - rng.uniform(0, 40, len(kc)), -1).astype(np.float32)
    start, items = orc.assign_grid(kc, BOUNDS)
    return dict(uv=uv, inv_z=inv_z, last_octave=kl["octave"][pick].astype(np.int32),
                last_angle=kl["angle"][pick].astype(np.float32), desc_mp=desc_mp, has_obs=has_obs, kc=kc, dc=dc,
                start=start, items=items, sf=sf, th=th, mbf=mbf, direction=direction, u_right=u_right,
                occupied=occupied)


def run_both(c, check_ori=True, occupied=True):
    hm = m.HammingMatcher()
    occ = c["occupied"] if occupied else None
    want = orc.search_by_projection_frame(c["uv"], c["inv_z"], c["last_octave"], c["last_angle"], c["desc_mp"],
                                          c["has_obs"], c["kc"], c["dc"], BOUNDS, c["sf"], c["th"], c["mbf"],
                                          c["direction"], c["u_right"], occ, check_ori)
    got = hm.search_by_projection_frame(c["uv"], c["inv_z"], c["last_octave"], c["last_angle"], c["desc_mp"],
                                        c["has_obs"], c["kc"], c["dc"], c["start"], c["items"], BOUNDS, c["sf"],
                                        c["th"], c["mbf"], c["direction"], c["u_right"], occ, check_ori)
    assert got[0] == want[0], "nmatches %d vs %d" % (got[0], want[0])
    assert np.array_equal(got[1], want[1])
    return got


@pytest.mark.parametrize("seed,th,direction,stereo", [(1, 15, 0, False), (2, 30, 0, False), (3, 7, 0, True),
                                                       (4, 14, 1, True), (5, 14, 2, True), (6, 100, 0, True)])
def test_search_by_projection_frame(seed, th, direction, stereo):
    c = make_case(seed, th, direction, stereo)
    n, matches = run_both(c)
    assert n > 30
    run_both(c, check_ori=False)
    run_both(c, occupied=False)


@pytest.mark.parametrize("seed,th,ratio,stereo", [(11, 1.0, 0.8, False), (12, 3.0, 0.8, True), (13, 5.0, 0.8, True),
                                                   (14, 1.0, 0.6, True)])
def test_search_by_projection_points(seed, th, ratio, stereo):
    """The local-map overload (src/ORBmatcher.cc:41-123): radius from the viewing angle, levels l-1..l, ratio test
    only between candidates of one level."""
    c = make_case(seed, 15, 0, stereo)
    rng = np.random.default_rng(seed)
    npts = len(c["uv"])
    view_cos = rng.uniform(0.99, 1.0, npts).astype(np.float32)
    level = np.clip(c["last_octave"] + rng.integers(-1, 2, npts), 0, 7).astype(np.int32)
    uv = (c["uv"] + rng.normal(0, 1.0, (npts, 2))).astype(np.float32)
    xr = (uv[:, 0] - rng.uniform(0, 40, npts)).astype(np.float32) if stereo else None
    hm = m.HammingMatcher()
    for occ in (c["occupied"], None):
        want = orc.search_by_projection_points(uv, view_cos, level, c["desc_mp"], c["has_obs"], c["kc"], c["dc"], BOUNDS,
                                               c["sf"], th, ratio, xr, c["u_right"], occ)
        got = hm.search_by_projection_points(uv, view_cos, level, c["desc_mp"], c["has_obs"], c["kc"], c["dc"],
                                             c["start"], c["items"], BOUNDS, c["sf"], th, ratio, xr, c["u_right"], occ)
        assert got[0] == want[0] and np.array_equal(got[1], want[1])
    if th >= 3:
        assert got[0] > 20


def test_relocalisation_overload_through_the_same_entry():
    """SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (src/ORBmatcher.cc:1518-1650): threshold
    ORBdist, every held feature blocked, every assignment blocking, no test on invzc, no stereo test."""
    c = make_case(21, 10, 0, False)
    hm = m.HammingMatcher()
    npts = len(c["uv"])
    ones = np.ones(npts, np.uint8)
    invz = np.ones(npts, np.float32)
    for th, orb_dist in ((10, 100), (3, 64)):
        want = orc.search_by_projection_frame(c["uv"], invz, c["last_octave"], c["last_angle"], c["desc_mp"], ones,
                                              c["kc"], c["dc"], BOUNDS, c["sf"], th, 0.0, 0, None, c["occupied"], True,
                                              orb_dist)
        got = hm.search_by_projection_frame(c["uv"], invz, c["last_octave"], c["last_angle"], c["desc_mp"], ones,
                                            c["kc"], c["dc"], c["start"], c["items"], BOUNDS, c["sf"], th, 0.0, 0, None,
                                            c["occupied"], True, orb_dist)
        assert got[0] == want[0] and np.array_equal(got[1], want[1])
    assert got[0] > 10


def test_search_by_projection_frame_edges():
    c = make_case(7, 15, 0, False, nfeatures=300)
    hm = m.HammingMatcher()
    args = lambda **kw: {**dict(proj_uv=c["uv"], inv_z=c["inv_z"], last_octave=c["last_octave"],
                                last_angle=c["last_angle"], desc_mp=c["desc_mp"], has_observations=c["has_obs"],
                                keys_un=c["kc"], desc=c["dc"], cell_start=c["start"], cell_items=c["items"],
                                bounds=BOUNDS, scale_factors=c["sf"], th=15.0), **kw}
    # no map points
    n, got = hm.search_by_projection_frame(**args(proj_uv=c["uv"][:0], inv_z=c["inv_z"][:0],
                                                  last_octave=c["last_octave"][:0], last_angle=c["last_angle"][:0],
                                                  desc_mp=c["desc_mp"][:0], has_observations=c["has_obs"][:0]))
    assert n == 0 and (got == -1).all() and len(got) == len(c["kc"])
    # everything behind the camera
    n, got = hm.search_by_projection_frame(**args(inv_z=-np.abs(c["inv_z"])))
    assert n == 0 and (got == -1).all()
    # an octave outside the scale table is skipped, not followed
    bad = c["last_octave"].copy()
    bad[:] = 99
    n, got = hm.search_by_projection_frame(**args(last_octave=bad))
    assert n == 0
    # a malformed grid is refused
    items = c["items"].copy()
    items[0] = len(c["kc"]) + 3
    with pytest.raises(m.OrbxError):
        hm.search_by_projection_frame(**args(cell_items=items))
