"""The allocation-free kNN entry point (orbx_hamming_knn2_noalloc_device, include/orbx.h) against
orbx_hamming_knn2_device: same (idx, d1, d2), bit for bit, on two calls in a row that share the library's
fixed partial-result array, the second with a smaller input.  Also checked against the CPU oracle on a
sample, and that problems beyond the array's size are refused without touching the outputs."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(hm, torch, q, db, noalloc, init=256, base=0):
    nq = q.shape[0]
    idx = torch.full((nq,), -7, dtype=torch.int64, device="cuda")
    d1 = torch.full((nq,), -7, dtype=torch.int32, device="cuda")
    d2 = torch.full((nq,), -7, dtype=torch.int32, device="cuda")
    hm.knn2_device(q, db, idx, d1, d2, init=init, index_base=base, noalloc=noalloc)
    hm.synchronize()
    return idx.cpu().numpy(), d1.cpu().numpy(), d2.cpu().numpy()


def test_noalloc_matches_allocating_entry_point_twice():
    import torch
    import my_orb_slam2_amd as m
    from tests import oracle as orc

    hm = m.HammingMatcher(0)
    g = torch.Generator(device="cuda").manual_seed(7)
    # (nq, ndb): the first call uses many chunks of the fixed array, the second (smaller) call leaves most of
    # what the first wrote in place; then many queries over a short database (one chunk, 300000 partials) and
    # tiny / degenerate shapes.
    cases = [(5000, 300000), (1237, 70001), (300000, 40000), (1, 1), (3, 0), (257, 2), (64, 5000)]
    for k, (nq, ndb) in enumerate(cases):
        db = torch.randint(0, 256, (max(ndb, 1), 32), dtype=torch.uint8, device="cuda", generator=g)[:ndb]
        q = torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device="cuda", generator=g)
        if ndb > 4:
            # planted near-duplicates and exact ties, so the best / second-best order matters
            src = torch.randint(0, ndb, (nq // 2,), device="cuda", generator=g)
            q[: nq // 2] = db[src]
            q[: nq // 4, 0] ^= 1
            db[ndb // 2] = db[ndb // 3]
        init = 256 if k % 2 == 0 else 2 ** 31 - 1
        base = 0 if k % 2 == 0 else 123456789012
        a = _run(hm, torch, q, db, noalloc=False, init=init, base=base)
        b = _run(hm, torch, q, db, noalloc=True, init=init, base=base)
        for x, y, name in zip(a, b, ("idx", "d1", "d2")):
            assert np.array_equal(x, y), (nq, ndb, name)
        if ndb and nq * ndb <= 5000 * 300000:
            n = min(nq, 64)
            oi, o1, o2 = orc.knn2(q[:n].cpu().numpy(), db.cpu().numpy(), init)
            oi = oi.astype(np.int64)
            oi = np.where(oi >= 0, oi + base, oi)
            assert np.array_equal(b[0][:n], oi) and np.array_equal(b[1][:n], o1) and np.array_equal(b[2][:n], o2)
    hm.close()


def test_noalloc_refuses_oversized_problem():
    import torch
    import my_orb_slam2_amd as m
    from my_orb_slam2_amd._lib import OrbxError

    hm = m.HammingMatcher(0)
    nq = 2 * 1024 * 1024 + 1                      # more queries than the array has partial results
    q = torch.zeros((nq, 32), dtype=torch.uint8, device="cuda")
    db = torch.zeros((8, 32), dtype=torch.uint8, device="cuda")
    idx = torch.full((nq,), -7, dtype=torch.int64, device="cuda")
    d1 = torch.full((nq,), -7, dtype=torch.int32, device="cuda")
    d2 = torch.full((nq,), -7, dtype=torch.int32, device="cuda")
    with pytest.raises(OrbxError) as ei:
        hm.knn2_device(q, db, idx, d1, d2, noalloc=True)
    assert ei.value.code == -3 and "fixed partial-result array" in str(ei.value)
    torch.cuda.synchronize()
    assert int((idx != -7).sum()) == 0 and int((d1 != -7).sum()) == 0
    # the allocating entry point takes the same problem
    hm.knn2_device(q, db, idx, d1, d2)
    hm.synchronize()
    assert int((d1 != 0).sum()) == 0 and int((idx != 0).sum()) == 0
    # fewer queries than the array has entries, but a database of three chunks: 3 * nq partials do not fit either
    nq = 1_000_000
    db = torch.zeros((9_000_000, 32), dtype=torch.uint8, device="cuda")
    idx.fill_(-7)
    with pytest.raises(OrbxError) as ei:
        hm.knn2_device(q[:nq], db, idx[:nq], d1[:nq], d2[:nq], noalloc=True)
    assert ei.value.code == -3
    torch.cuda.synchronize()
    assert int((idx != -7).sum()) == 0
    hm.close()
