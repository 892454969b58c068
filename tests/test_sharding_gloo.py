"""World-size-2 rehearsal of the multi-GPU paths on CPU (gloo): frame ownership needs no collective;
the sharded kNN exchanges one packed 16-byte record per query and rank in ONE all-gather.  Per-shard scans
come from the oracle here (no GPU in this test); the records are packed, gathered and unpacked by the
product's sharding module and the merged result must equal the oracle's scan of the whole database.  The
product's device-side fold of the gathered layout is covered by tests/test_gpu_match.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def fold(idx, d1, d2, init):
    """numpy statement of the merge rule (shards in row order)."""
    nparts, nq = idx.shape
    b1 = np.full(nq, init, np.int64)
    b2 = np.full(nq, init, np.int64)
    bi = np.full(nq, -1, np.int64)
    for p in range(nparts):
        a1, a2, ai = d1[p].astype(np.int64), d2[p].astype(np.int64), idx[p]
        ok = ai >= 0
        better = ok & (a1 < b1)
        second = ok & ~better & (a1 < b2)
        b2 = np.where(better, b1, np.where(second, a1, b2))
        bi = np.where(better, ai, bi)
        b1 = np.where(better, a1, b1)
        b2 = np.where(a2 < b2, a2, b2)
    return bi, b1, b2


def _worker(rank, world, port, q_np, db_np):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from my_orb_slam2_amd import sharding
    from tests import oracle as orc
    lo, hi = sharding.shard_range(len(db_np), rank, world)
    idx, d1, d2 = orc.knn2(q_np, db_np[lo:hi])
    idx = np.where(idx >= 0, idx.astype(np.int64) + lo, -1)
    hits = sharding.pack_hits(torch.from_numpy(idx), torch.from_numpy(d1), torch.from_numpy(d2))
    assert hits.shape == (len(q_np), 2) and hits.numpy().tobytes()[:16] == (
        np.int32(d1[0]).tobytes() + np.int32(d2[0]).tobytes() + np.int64(idx[0]).tobytes())   # orbx_knn2_hit layout
    gathered = sharding.all_gather_hits(hits)                     # one collective
    assert gathered.shape == (world, len(q_np), 2)
    gi, g1, g2 = sharding.unpack_hits(gathered)
    bi, b1, b2 = fold(gi.numpy(), g1.numpy(), g2.numpy(), 256)
    oi, o1, o2 = orc.knn2(q_np, db_np)
    ok = np.array_equal(bi, oi) and np.array_equal(b1, o1) and np.array_equal(b2, o2)
    # frames: every frame has exactly one owner, left/right of a pair share it
    mine = sharding.frames_of_rank(37, rank, world)
    counts = torch.zeros(37, dtype=torch.int32)
    counts[mine] = 1
    dist.all_reduce(counts)
    ok = ok and bool((counts == 1).all()) and all(sharding.frame_owner(k, world) == rank for k in mine)
    dist.destroy_process_group()
    assert ok, "rank %d: merged result or frame ownership differs" % rank


def test_sharded_knn_and_frame_ownership_world2():
    rng = np.random.default_rng(3)
    db = rng.integers(0, 256, (3001, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    q[:100] = db[rng.integers(0, 3001, 100)]
    q[:100, 0] ^= 3
    db[2500] = db[10]                      # duplicate rows in different shards: lowest index must win
    q[0] = db[10]
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), q, db), nprocs=world, join=True)   # raises if a rank fails


def test_shard_bounds():
    from my_orb_slam2_amd import sharding
    assert sharding.shard_bounds(10, 3) == [0, 4, 7, 10]
    assert sharding.shard_bounds(100_000_000, 8)[-1] == 100_000_000
    assert sharding.shard_range(5, 4, 8) == (4, 5) and sharding.shard_range(5, 7, 8) == (5, 5)
    assert [sharding.frame_owner(k, 8) for k in (0, 7, 8, 13)] == [0, 7, 0, 5]
