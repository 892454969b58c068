#!/usr/bin/env python3
"""BASELINE.json configs[4] at its full size: 1 M query descriptors against a 100 M row database, top-2 Hamming
search + ratio test, database split into contiguous shards, per-shard results merged exactly.

    python tools/knn_full.py                         # one GPU walks the 8 shards one after the other
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/knn_full.py
                                                     # N ranks, 8/N shards each, one all-gather (RCCL) + merge

Data comes from a counter-based splitmix64, so any database row can be made from its index alone: row i is the
four 64-bit outputs for counters 4i..4i+3 (seed 1). The first half of the queries are planted: query q copies
database row src(q) and flips bits at r(q) in [0, 40] pseudo-random positions; the second half are uniform.

Checks (no CPU scan of 10^14 pairs is possible, so size-independent properties plus an independent sample):
  * every planted query must come back with idx == src(q) and d1 == popcount(flip mask);
  * the ratio test (TH_LOW 50, ratio 0.6) must accept exactly the planted queries;
  * d1 <= d2 everywhere and idx inside the database;
  * `--sample` queries are recomputed against the whole database with plain torch ops (xor, byte popcount table,
    min over (distance, index) keys) and must agree in idx, d1 and d2.
Prints one JSON object on rank 0 and exits non-zero on any disagreement.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GOLD = 0x9E3779B97F4A7C15
M1 = 0xBF58476D1CE4E5B9
M2 = 0x94D049BB133111EB


def s64(v):
    """Python int -> the int64 with the same low 64 bits."""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def splitmix64(torch, counter, seed):
    """Output of splitmix64 number `counter` (int64 tensor, taken as unsigned) of the stream seeded `seed`."""
    def lsr(x, k):
        return (x >> k) & ((1 << (64 - k)) - 1)
    z = (counter + 1) * s64(GOLD) + s64(seed)
    z = (z ^ lsr(z, 30)) * s64(M1)
    z = (z ^ lsr(z, 27)) * s64(M2)
    return z ^ lsr(z, 31)


def db_rows(torch, first, count, device):
    """Database rows first..first+count-1 as a (count, 32) uint8 tensor."""
    c = torch.arange(first * 4, (first + count) * 4, dtype=torch.int64, device=device)
    return splitmix64(torch, c, 1).view(torch.uint8).view(count, 32)


def rows_by_index(torch, index):
    c = (index[:, None] * 4 + torch.arange(4, dtype=torch.int64, device=index.device)[None, :]).reshape(-1)
    return splitmix64(torch, c, 1).view(torch.uint8).view(index.shape[0], 32)


def make_queries(torch, nq, ndb, device):
    """-> queries (nq, 32) u8, src (nq/2,) i64, planted distance (nq/2,) i32"""
    half = nq // 2
    qid = torch.arange(half, dtype=torch.int64, device=device)
    h = splitmix64(torch, qid, 2)
    src = (h & ((1 << 62) - 1)) % ndb
    r = (splitmix64(torch, qid, 3) & 0xFFFF) % 41
    mask = torch.zeros((half, 32), dtype=torch.uint8, device=device)
    flat = mask.view(-1)
    for k in range(40):
        bit = splitmix64(torch, qid * 64 + k, 4) & 255
        on = k < r
        pos = (qid * 32 + (bit >> 3))[on]
        val = (torch.ones_like(bit) << (bit & 7)).to(torch.uint8)[on]
        flat[pos] = flat[pos] ^ val                   # one write per query per pass: no index collides
    lut = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int32, device=device)
    planted = lut[mask.to(torch.int64)].sum(dim=1).to(torch.int32)
    q = torch.empty((nq, 32), dtype=torch.uint8, device=device)
    q[:half] = rows_by_index(torch, src) ^ mask
    c = torch.arange((nq - half) * 4, dtype=torch.int64, device=device)
    q[half:] = splitmix64(torch, c, 5).view(torch.uint8).view(nq - half, 32)
    return q, src, planted


def torch_top2(torch, q_row, db, base, lut):
    """Best two of one query over a database shard with plain torch ops: (key1, key2), key = d << 40 | global index."""
    d = lut[(db ^ q_row[None, :]).to(torch.int64)].sum(dim=1)
    key = (d.to(torch.int64) << 40) | (torch.arange(db.shape[0], dtype=torch.int64, device=db.device) + base)
    two = torch.topk(key, 2, largest=False).values
    return int(two[0].item()), int(two[1].item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nq", type=int, default=1_000_000)
    ap.add_argument("--ndb", type=int, default=100_000_000)
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--slice", type=int, default=131072, help="queries per kernel launch")
    ap.add_argument("--sample", type=int, default=16, help="queries recomputed with torch ops over the whole database")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    import my_orb_slam2_amd as m
    from my_orb_slam2_amd.sharding import shard_bounds, all_gather_hits, pack_hits

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    nshards = max(a.shards, world)
    if nshards % world:
        raise SystemExit("--shards must be a multiple of the number of ranks")
    per = nshards // world
    bounds = shard_bounds(a.ndb, nshards)
    mine = list(range(rank * per, (rank + 1) * per))          # consecutive shards: folding them in order is exact

    q, src, planted = make_queries(torch, a.nq, a.ndb, dev)
    dbs = [db_rows(torch, bounds[s], bounds[s + 1] - bounds[s], dev) for s in mine]
    hm = m.HammingMatcher(local)
    nq = a.nq
    pi = torch.empty((per, nq), dtype=torch.int64, device=dev)
    p1 = torch.empty((per, nq), dtype=torch.int32, device=dev)
    p2 = torch.empty((per, nq), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k, s in enumerate(mine):
        for s0 in range(0, nq, a.slice):
            s1 = min(nq, s0 + a.slice)
            hm.knn2_device(q[s0:s1], dbs[k], pi[k, s0:s1], p1[k, s0:s1], p2[k, s0:s1], index_base=bounds[s])
        hm.synchronize()
        if rank == 0:
            print("shard %d/%d done, %.1f s" % (k + 1, per, time.perf_counter() - t0), file=sys.stderr, flush=True)
    li, l1, l2 = (torch.empty(nq, dtype=t.dtype, device=dev) for t in (pi, p1, p2))
    hm.merge_device(pi, p1, p2, li, l1, l2)
    hm.synchronize()
    gathered = all_gather_hits(pack_hits(li, l1, l2))          # one collective of nq * 16 bytes per rank
    idx, d1, d2 = torch.empty_like(li), torch.empty_like(l1), torch.empty_like(l2)
    hm.merge_packed_device(gathered, gathered.shape[0], idx, d1, d2)
    acc = torch.empty(nq, dtype=torch.uint8, device=dev)
    hm.ratio_test_device(d1, d2, acc, th=50, inclusive=True, ratio=0.6)
    hm.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0

    half = nq // 2
    st = {
        "planted_queries": half,
        "planted_idx_wrong": int((idx[:half] != src).sum().item()),
        "planted_d1_wrong": int((d1[:half] != planted).sum().item()),
        "accepted": int(acc.sum().item()),
        "accepted_not_planted": int(acc[half:].sum().item()),
        "planted_not_accepted": int((acc[:half] == 0).sum().item()),
        "d1_gt_d2": int((d1 > d2).sum().item()),
        "idx_out_of_range": int(((idx < 0) | (idx >= a.ndb)).sum().item()),
    }
    # independent recomputation of a few queries: every rank scans its shards, keys are min-reduced over ranks
    lut = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int32, device=dev)
    ns = min(a.sample, nq)
    picks = [(i * (nq // ns) + (i % 2) * half) % nq for i in range(ns)]    # planted and uniform ones alternate
    wrong = 0
    for qi in picks:
        keys = []
        for k, s in enumerate(mine):
            keys.extend(torch_top2(torch, q[qi], dbs[k], bounds[s], lut))
        kt = torch.tensor(sorted(keys)[:2], dtype=torch.int64, device=dev)
        if world > 1:
            allk = [torch.empty_like(kt) for _ in range(world)]
            dist.all_gather(allk, kt)
            kt = torch.sort(torch.cat(allk)).values[:2]
        k1, k2 = int(kt[0].item()), int(kt[1].item())
        got = (int(idx[qi].item()), int(d1[qi].item()), int(d2[qi].item()))
        if got != (k1 & ((1 << 40) - 1), k1 >> 40, k2 >> 40):
            wrong += 1
    st["sampled_queries"] = ns
    st["sampled_wrong"] = wrong
    bad = sum(st[k] for k in ("planted_idx_wrong", "planted_d1_wrong", "accepted_not_planted", "planted_not_accepted",
                              "d1_gt_d2", "idx_out_of_range", "sampled_wrong")) + (st["accepted"] != half)
    if rank == 0:
        print(json.dumps({"workload": "configs[4] full size: %d queries x %d database rows, %d shards, top-2 + merge + "
                                      "ratio test" % (nq, a.ndb, nshards),
                          "n_gpus": world, "seconds": round(dt, 2), "value": nq * a.ndb / dt, "unit": "Hamming pairs/s",
                          "collective": "one all_gather of nq x 16 B per rank" if world > 1 else "none (one GPU walks the shards)",
                          "checks": st}, indent=1))
    if world > 1:
        dist.destroy_process_group()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
