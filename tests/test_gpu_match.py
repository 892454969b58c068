"""GPU parity tests for the Hamming matcher: match lists must be bit-exact against the oracle."""
import numpy as np
import pytest

import my_orb_slam2_amd as m
from tests import oracle as orc

pytestmark = pytest.mark.gpu


def planted(rng, nq, ndb):
    db = rng.integers(0, 256, (ndb, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    half = nq // 2
    src = rng.integers(0, ndb, half)
    q[:half] = db[src]
    for i in range(half):                          # flip r in [0, 40] random bits
        for b in rng.integers(0, 256, rng.integers(0, 41)):
            q[i, b // 8] ^= np.uint8(1 << (b % 8))
    return q, db


@pytest.mark.parametrize("nq,ndb", [(1, 1), (7, 0), (5, 1), (300, 1000), (1000, 5000), (513, 40000), (64, 300000)])
def test_knn2_matches_oracle(nq, ndb):
    rng = np.random.default_rng(nq * 7919 + ndb)
    q, db = planted(rng, nq, max(ndb, 1))
    db = db[:ndb]
    hm = m.HammingMatcher()
    for init in (256, 2**31 - 1):
        idx, d1, d2 = hm.knn2(q, db, init=init)
        oi, o1, o2 = orc.knn2(q, db, init=init)
        assert np.array_equal(idx, oi) and np.array_equal(d1, o1) and np.array_equal(d2, o2)


@pytest.mark.parametrize("nq,ndb,blocks", [(4000, 600000, 2), (33000, 600000, 8)])
def test_knn2_scan_plans_with_more_queries_per_wave(nq, ndb, blocks):
    """Sizes at which the scan keeps two and eight blocks of 32 queries per wave (the small cases above all run with
    one): a wave's operand registers, the merge of a query's two half-lanes and the chunk fold are the same code,
    but unrolled differently. A sample of the queries is checked against the oracle, all of them for consistency."""
    rng = np.random.default_rng(nq + blocks)
    q, db = planted(rng, 512, ndb)
    q = np.concatenate([q, rng.integers(0, 256, (nq - 512, 32), dtype=np.uint8)])
    db[ndb // 3] = db[ndb // 7]                            # an exact duplicate: lowest row must win
    q[5] = db[ndb // 3]
    q[nq - 1] = db[ndb - 1]                                # last row of the last chunk, last query of the last wave
    hm = m.HammingMatcher()
    idx, d1, d2 = hm.knn2(q, db)
    pick = np.concatenate([np.arange(0, 96), np.arange(nq - 32, nq)])
    oi, o1, o2 = orc.knn2(q[pick], db)
    assert np.array_equal(idx[pick], oi) and np.array_equal(d1[pick], o1) and np.array_equal(d2[pick], o2)
    assert idx[5] == ndb // 7 and d1[5] == 0 and d2[5] == 0
    assert idx[nq - 1] == ndb - 1 and d1[nq - 1] == 0
    assert np.all(d1 <= d2) and np.all(idx >= 0) and np.all(idx < ndb)
    # the database rows really are what came back
    rows = db[idx[:2000]]
    dist = np.unpackbits(rows ^ q[:2000], axis=1).sum(axis=1)
    assert np.array_equal(dist, d1[:2000])


def test_knn2_ties_and_duplicates():
    rng = np.random.default_rng(9)
    base = rng.integers(0, 256, (50, 32), dtype=np.uint8)
    db = np.concatenate([base, base, base[::-1]])          # every row appears three times
    q = base.copy()
    q[::2, 0] ^= 1
    hm = m.HammingMatcher()
    idx, d1, d2 = hm.knn2(q, db)
    oi, o1, o2 = orc.knn2(q, db)
    assert np.array_equal(idx, oi) and np.array_equal(d1, o1) and np.array_equal(d2, o2)
    assert np.array_equal(d1, d2)                          # a tie with the best sets second = best
    assert np.all(idx < 50)                                # lowest index wins


def test_knn2_monotone_orders_and_short_tails():
    """Database rows ordered so that the best two change at every row (distances falling), never after the
    first two (rising), or stay equal; lengths around the kernel's row grouping."""
    rng = np.random.default_rng(21)
    q = rng.integers(0, 256, (70, 32), dtype=np.uint8)
    target = q[0].copy()
    rows = []
    for nbits in range(0, 250):                            # row k differs from q[0] in exactly k bits
        r = target.copy()
        for b in range(nbits):
            r[b // 8] ^= np.uint8(1 << (b % 8))
        rows.append(r)
    rising = np.stack(rows)
    hm = m.HammingMatcher()
    for db in (rising, rising[::-1].copy(), np.repeat(rising[40:41], 37, axis=0)):
        for n in (len(db), 2, 3, 4, 5, 6, 7, 9, 37):
            part = np.ascontiguousarray(db[:n])
            idx, d1, d2 = hm.knn2(q, part)
            oi, o1, o2 = orc.knn2(q, part)
            assert np.array_equal(idx, oi) and np.array_equal(d1, o1) and np.array_equal(d2, o2)


def test_scalar_distance():
    rng = np.random.default_rng(10)
    for _ in range(50):
        a = rng.integers(0, 256, 32, dtype=np.uint8)
        b = rng.integers(0, 256, 32, dtype=np.uint8)
        assert m.ORBmatcher.DescriptorDistance(a, b) == orc.hamming(a, b)


def test_sharded_merge_and_ratio_test():
    import torch
    rng = np.random.default_rng(11)
    nq, ndb, shards = 2000, 30000, 4
    q, db = planted(rng, nq, ndb)
    hm = m.HammingMatcher()
    dq = torch.from_numpy(q).cuda()
    part_idx = torch.empty((shards, nq), dtype=torch.int64, device="cuda")
    part_d1 = torch.empty((shards, nq), dtype=torch.int32, device="cuda")
    part_d2 = torch.empty((shards, nq), dtype=torch.int32, device="cuda")
    bounds = np.linspace(0, ndb, shards + 1).astype(int)
    keep = []
    for s in range(shards):
        ddb = torch.from_numpy(db[bounds[s]:bounds[s + 1]]).cuda()
        keep.append(ddb)
        hm.knn2_device(dq, ddb, part_idx[s], part_d1[s], part_d2[s], index_base=int(bounds[s]))
    o_idx = torch.empty(nq, dtype=torch.int64, device="cuda")
    o_d1 = torch.empty(nq, dtype=torch.int32, device="cuda")
    o_d2 = torch.empty(nq, dtype=torch.int32, device="cuda")
    hm.merge_device(part_idx, part_d1, part_d2, o_idx, o_d1, o_d2)
    acc = torch.empty(nq, dtype=torch.uint8, device="cuda")
    hm.ratio_test_device(o_d1, o_d2, acc, th=m.TH_LOW, inclusive=True, ratio=0.6)
    hm.synchronize()
    oi, o1, o2 = orc.knn2(q, db)
    assert np.array_equal(o_idx.cpu().numpy(), oi)
    assert np.array_equal(o_d1.cpu().numpy(), o1) and np.array_equal(o_d2.cpu().numpy(), o2)
    want = np.array([orc.lib().orc_ratio_accept(int(a), int(b), 50, 1, 0.6) for a, b in zip(o1, o2)], np.uint8)
    assert np.array_equal(acc.cpu().numpy(), want)
    assert 0.3 < want.mean() < 0.6                          # planted half passes, random half fails


def test_stereo_band_match_on_extracted_pairs():
    import torch
    npairs = 3
    ex_l = m.ORBextractor(1200, 1.2, 8, 20, 7)
    ex_r = m.ORBextractor(1200, 1.2, 8, 20, 7)
    cap = ex_l.capacity
    kl = np.zeros((npairs, cap), m.KEYPOINT_DTYPE)
    kr = np.zeros((npairs, cap), m.KEYPOINT_DTYPE)
    dl = np.zeros((npairs, cap, 32), np.uint8)
    dr = np.zeros((npairs, cap, 32), np.uint8)
    nl = np.zeros(npairs, np.int32)
    nr = np.zeros(npairs, np.int32)
    for p in range(npairs):
        a, b = ex_l(m.synth_frame(40 + p))
        c, d = ex_r(m.synth_frame(40 + p, right_view=True))
        nl[p], nr[p] = len(a), len(c)
        kl[p, :len(a)], dl[p, :len(a)] = a, b
        kr[p, :len(c)], dr[p, :len(c)] = c, d
    scale = ex_l.GetScaleFactors()
    bf, b = 47.90639384423901, 47.90639384423901 / 435.2046959714599    # EuRoC stereo: Camera.bf, bf / fx
    hm = m.HammingMatcher()
    t = lambda a: torch.from_numpy(a.view(np.uint8) if a.dtype == m.KEYPOINT_DTYPE else a).cuda()
    best_idx = torch.empty((npairs, cap), dtype=torch.int32, device="cuda")
    best_dist = torch.empty((npairs, cap), dtype=torch.int32, device="cuda")
    hm.stereo_band_match_device(t(kl), t(dl), t(nl), t(kr), t(dr), t(nr), scale, bf, b, best_idx, best_dist)
    hm.synchronize()
    bi, bd = best_idx.cpu().numpy(), best_dist.cpu().numpy()
    matched = 0
    for p in range(npairs):
        oi, od = orc.stereo_band_match(kl[p, :nl[p]], dl[p, :nl[p]], kr[p, :nr[p]], dr[p, :nr[p]], 480, scale, bf, b)
        assert np.array_equal(bi[p, :nl[p]], oi) and np.array_equal(bd[p, :nl[p]], od)
        matched += int((od < 75).sum())
    assert matched > 300


@pytest.mark.parametrize("nfeatures,npairs,min_total", [(1200, 4, 400), (3000, 2, 300), (60, 3, 5)])
def test_compute_stereo_matches_full(nfeatures, npairs, min_total):
    """Band search + SAD refinement + median cut against the oracle's Frame::ComputeStereoMatches; also with more
    keypoints per image than the median kernel keeps in registers, and with very few."""
    import torch
    ex_l = m.ORBextractor(nfeatures, 1.2, 8, 20, 7, max_batch=npairs)
    ex_r = m.ORBextractor(nfeatures, 1.2, 8, 20, 7, max_batch=npairs)
    cap = ex_l.capacity
    left = np.stack([m.synth_frame(60 + p) for p in range(npairs)])
    right = np.stack([m.synth_frame(60 + p, right_view=True) for p in range(npairs)])
    d_left, d_right = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()

    def alloc():
        return (torch.zeros((npairs, cap, 7), dtype=torch.float32, device="cuda"),
                torch.zeros((npairs, cap, 32), dtype=torch.uint8, device="cuda"),
                torch.zeros(npairs, dtype=torch.int32, device="cuda"))
    kl, dl, nl = alloc()
    kr, dr, nr = alloc()
    ex_l.extract_batch_device(d_left, kl, dl, nl)
    ex_r.extract_batch_device(d_right, kr, dr, nr)
    bf, b = 47.90639384423901, 47.90639384423901 / 435.2046959714599
    ur = torch.empty((npairs, cap), dtype=torch.float32, device="cuda")
    z = torch.empty((npairs, cap), dtype=torch.float32, device="cuda")
    hm = m.HammingMatcher()
    hm.compute_stereo_matches_device(ex_l, ex_r, kl, dl, nl, kr, dr, nr, bf, b, ur, z)
    hm.synchronize()
    ur, z = ur.cpu().numpy(), z.cpu().numpy()
    nl = nl.cpu().numpy()
    total = 0
    for p in range(npairs):
        ol, orr = orc.Extractor(nfeatures), orc.Extractor(nfeatures)
        okl, odl = ol.extract(left[p])
        okr, odr = orr.extract(right[p])
        assert nl[p] == len(okl)
        our, oz, kept = orc.compute_stereo_matches(ol, orr, okl, odl, okr, odr, bf, b)
        assert np.array_equal(ur[p, :nl[p]], our), "uRight differs in pair %d" % p
        assert np.array_equal(z[p, :nl[p]], oz), "depth differs in pair %d" % p
        assert kept == int((our >= 0).sum())
        total += kept
    if nfeatures == 3000:
        assert nl.max() > 2048             # beyond the register-resident path of the median kernel
    assert total > min_total


def test_stereo_then_extraction_on_another_stream():
    """A context may be driven from different streams in turn. Extraction on stream X, ComputeStereoMatches on the
    matcher's stream, then at once an extraction of other frames on stream Y with the same two contexts: the second
    extraction overwrites the pyramids the SAD refinement reads, so it has to start behind the stereo kernels. The
    stereo result must equal the one computed with nothing queued behind it."""
    import torch
    npairs, nfeatures = 16, 1200
    ex_l = m.ORBextractor(nfeatures, 1.2, 8, 20, 7, max_batch=npairs)
    ex_r = m.ORBextractor(nfeatures, 1.2, 8, 20, 7, max_batch=npairs)
    cap = ex_l.capacity
    left = np.stack([m.synth_frame(300 + p) for p in range(npairs)])
    right = np.stack([m.synth_frame(300 + p, right_view=True) for p in range(npairs)])
    rng = np.random.default_rng(7)
    other = rng.integers(0, 256, left.shape, dtype=np.uint8)
    d_left, d_right, d_other = (torch.from_numpy(a).cuda() for a in (left, right, other))

    def alloc():
        return (torch.zeros((npairs, cap, 7), dtype=torch.float32, device="cuda"),
                torch.zeros((npairs, cap, 32), dtype=torch.uint8, device="cuda"),
                torch.zeros(npairs, dtype=torch.int32, device="cuda"))
    bf, b = 47.90639384423901, 47.90639384423901 / 435.2046959714599
    hm = m.HammingMatcher()
    sx, sy = torch.cuda.Stream(), torch.cuda.Stream()
    results = []
    for overlapped in (False, True):
        kl, dl, nl = alloc()
        kr, dr, nr = alloc()
        k2, d2, n2 = alloc()
        k3, d3, n3 = alloc()
        ur = torch.full((npairs, cap), -7.0, dtype=torch.float32, device="cuda")
        z = torch.full((npairs, cap), -7.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ex_l.extract_batch_device(d_left, kl, dl, nl, stream=sx.cuda_stream)
        ex_r.extract_batch_device(d_right, kr, dr, nr, stream=sx.cuda_stream)
        hm.compute_stereo_matches_device(ex_l, ex_r, kl, dl, nl, kr, dr, nr, bf, b, ur, z)
        if overlapped:
            ex_l.extract_batch_device(d_other, k2, d2, n2, stream=sy.cuda_stream)
            ex_r.extract_batch_device(d_other, k3, d3, n3, stream=sy.cuda_stream)
        hm.synchronize()
        torch.cuda.synchronize()
        results.append((ur.cpu().numpy(), z.cpu().numpy(), nl.cpu().numpy()))
    (ur0, z0, n0), (ur1, z1, n1) = results
    assert np.array_equal(n0, n1)
    assert np.array_equal(ur0, ur1) and np.array_equal(z0, z1)
    assert int((ur0 >= 0).sum()) > 50 * npairs
    # and the serial result is the oracle's
    ol, orr = orc.Extractor(nfeatures), orc.Extractor(nfeatures)
    okl, odl = ol.extract(left[0])
    okr, odr = orr.extract(right[0])
    our, oz, _ = orc.compute_stereo_matches(ol, orr, okl, odl, okr, odr, bf, b)
    assert np.array_equal(ur0[0, :n0[0]], our) and np.array_equal(z0[0, :n0[0]], oz)


def test_four_stereo_sets_in_flight():
    """Four independent sets (two extractors and a matcher each: twelve streams over the four hardware queues a
    process has) take turns for several rounds with no host synchronisation in between: extraction left, extraction
    right and ComputeStereoMatches of a set are ordered by events only. Every set sees the same pairs, so every
    set's result must equal the one a single set computes with a synchronisation after every call."""
    import torch
    npairs, nfeatures, nsets, rounds = 8, 1200, 4, 6
    left = np.stack([m.synth_frame(500 + p) for p in range(npairs)])
    right = np.stack([m.synth_frame(500 + p, right_view=True) for p in range(npairs)])
    d_left, d_right = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    bf, b = 47.90639384423901, 47.90639384423901 / 435.2046959714599

    def make_set():
        exl = m.ORBextractor(nfeatures, 1.2, 8, 20, 7, max_batch=npairs)
        exr = m.ORBextractor(nfeatures, 1.2, 8, 20, 7, max_batch=npairs)
        cap = exl.capacity

        def alloc():
            return (torch.zeros((npairs, cap, 7), dtype=torch.float32, device="cuda"),
                    torch.zeros((npairs, cap, 32), dtype=torch.uint8, device="cuda"),
                    torch.zeros(npairs, dtype=torch.int32, device="cuda"))
        return dict(exl=exl, exr=exr, L=alloc(), R=alloc(), hm=m.HammingMatcher(),
                    ur=torch.full((npairs, cap), -7.0, dtype=torch.float32, device="cuda"),
                    z=torch.full((npairs, cap), -7.0, dtype=torch.float32, device="cuda"))

    def step(s):
        s["exl"].extract_batch_device(d_left, *s["L"])
        s["exr"].extract_batch_device(d_right, *s["R"])
        s["hm"].compute_stereo_matches_device(s["exl"], s["exr"], *s["L"], *s["R"], bf, b, s["ur"], s["z"])

    ref = make_set()
    step(ref)
    ref["hm"].synchronize()
    torch.cuda.synchronize()
    want_ur, want_z = ref["ur"].cpu().numpy(), ref["z"].cpu().numpy()
    assert int((want_ur >= 0).sum()) > 50 * npairs
    sets = [make_set() for _ in range(nsets)]
    for r in range(rounds):
        for s in sets:
            step(s)
    for s in sets:
        s["hm"].synchronize()
    torch.cuda.synchronize()
    for k, s in enumerate(sets):
        assert np.array_equal(s["ur"].cpu().numpy(), want_ur), "set %d" % k
        assert np.array_equal(s["z"].cpu().numpy(), want_z), "set %d" % k


@pytest.mark.parametrize("nshards", [1, 2, 3, 8])
def test_packed_scan_and_merge_of_simulated_ranks(nshards):
    """The sharded-database path without a process group: every "rank" scans its contiguous shard into packed
    16-byte records, the records are laid out as one all-gather would leave them ([rank][query]) and the
    product's device-side fold must give the oracle's scan of the whole database, ties and duplicates included."""
    import torch
    from my_orb_slam2_amd import sharding
    rng = np.random.default_rng(100 + nshards)
    q, db = planted(rng, 700, 9001)
    db[8000] = db[17]                      # duplicates in different shards: the lowest row must win
    db[4500] = db[17]
    q[0] = db[17]
    q[1] = db[8000]
    q[1, 3] ^= 4
    hm = m.HammingMatcher()
    dq, ddb = torch.from_numpy(q).cuda(), torch.from_numpy(db).cuda()
    bounds = sharding.shard_bounds(len(db), nshards)
    for init in (256, 2**31 - 1):
        gathered = torch.empty((nshards, len(q), 2), dtype=torch.int64, device="cuda")
        for r in range(nshards):
            hm.knn2_packed_device(dq, ddb[bounds[r]:bounds[r + 1]], gathered[r], init=init, index_base=bounds[r])
        oi, o1, o2 = orc.knn2(q, db, init=init)
        # the packed records of one shard carry the three-array result of the same scan
        idx = torch.empty(len(q), dtype=torch.int64, device="cuda")
        d1 = torch.empty(len(q), dtype=torch.int32, device="cuda")
        d2 = torch.empty(len(q), dtype=torch.int32, device="cuda")
        hm.knn2_device(dq, ddb[bounds[0]:bounds[1]], idx, d1, d2, init=init, index_base=bounds[0])
        hm.synchronize()
        pi, p1, p2 = sharding.unpack_hits(gathered[0])
        assert torch.equal(pi, idx) and torch.equal(p1, d1) and torch.equal(p2, d2)
        assert torch.equal(sharding.pack_hits(idx, d1, d2), gathered[0])
        hm.merge_packed_device(gathered, nshards, idx, d1, d2, init=init)
        hm.synchronize()
        assert np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(d1.cpu().numpy(), o1)
        assert np.array_equal(d2.cpu().numpy(), o2)


def _sharded_worker(rank, world, port, q_np, db_np, out_dir):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from my_orb_slam2_amd import sharding
    lo, hi = sharding.shard_range(len(db_np), rank, world)
    sm = sharding.ShardedHammingMatcher(torch.from_numpy(db_np[lo:hi]).cuda(), lo, device=0)
    idx, d1, d2 = sm.knn2(torch.from_numpy(q_np).cuda(), slice_rows=300)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), idx=idx.cpu().numpy(), d1=d1.cpu().numpy(), d2=d2.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_matcher_two_processes_one_gpu(tmp_path):
    """ShardedHammingMatcher end to end with a real process group: two ranks share this box's one GPU (gloo
    moves the records, staged through the host; on a multi-GPU node the same code runs over RCCL)."""
    import socket
    import torch.multiprocessing as mp
    rng = np.random.default_rng(5)
    q, db = planted(rng, 1000, 20001)
    db[15000] = db[3]
    q[0] = db[3]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_sharded_worker, args=(2, port, q, db, str(tmp_path)), nprocs=2, join=True)
    oi, o1, o2 = orc.knn2(q, db)
    for r in range(2):
        z = np.load(str(tmp_path / ("r%d.npz" % r)))
        assert np.array_equal(z["idx"], oi) and np.array_equal(z["d1"], o1) and np.array_equal(z["d2"], o2)
