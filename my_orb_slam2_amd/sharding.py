"""One process per GPU: how frames, stereo pairs and a large descriptor database are split.

Extraction and stereo matching need no communication (frame / pair k belongs to rank k mod G).
The brute-force kNN over a sharded database has exactly one exchange step: every rank scans its
contiguous shard of database rows for all queries, the (idx, d1, d2) triples are all-gathered
(RCCL when the process group is "nccl"), and every rank folds them with orbx_knn2_merge_device.
The fold is exact: the two smallest distances of the whole database lie inside the union of the
per-shard best two, and shards are folded in row order so the lowest row index wins ties, as in
the sequential scan of my_orb_slam2/src/ORBmatcher.cc:204-229.
"""
import torch
import torch.distributed as dist

from .matcher import HammingMatcher


def shard_bounds(n, world):
    """Contiguous, balanced split of n rows into `world` shards: list of world+1 offsets."""
    base, rem = divmod(int(n), int(world))
    out = [0]
    for r in range(world):
        out.append(out[-1] + base + (1 if r < rem else 0))
    return out


def shard_range(n, rank, world):
    b = shard_bounds(n, world)
    return b[rank], b[rank + 1]


def frame_owner(k, world):
    """Frame (or stereo pair) k is extracted on rank k mod G; left and right of a pair stay together."""
    return int(k) % int(world)


def frames_of_rank(nframes, rank, world):
    return list(range(int(rank), int(nframes), int(world)))


def all_gather_partials(idx, d1, d2, group=None):
    """All-gather per-shard results -> (world, nq) tensors, shard order = rank order."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return idx[None], d1[None], d2[None]
    outs = []
    for t in (idx, d1, d2):
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t.contiguous(), group=group)
        outs.append(torch.stack(parts))
    return tuple(outs)


class ShardedHammingMatcher:
    """kNN top-2 against a database whose rows are split across the ranks of a process group."""

    def __init__(self, d_db_shard, index_base, device=0, group=None):
        """d_db_shard: this rank's (rows, 32) uint8 CUDA tensor; index_base: global index of its row 0."""
        self.m = HammingMatcher(device)
        self.db = d_db_shard
        self.base = int(index_base)
        self.group = group

    def knn2(self, d_query, init=256):
        nq = d_query.shape[0]
        dev = d_query.device
        idx = torch.empty(nq, dtype=torch.int64, device=dev)
        d1 = torch.empty(nq, dtype=torch.int32, device=dev)
        d2 = torch.empty(nq, dtype=torch.int32, device=dev)
        self.m.knn2_device(d_query, self.db, idx, d1, d2, init=init, index_base=self.base)
        self.m.synchronize()
        gi, g1, g2 = all_gather_partials(idx, d1, d2, self.group)
        o_idx, o_d1, o_d2 = torch.empty_like(idx), torch.empty_like(d1), torch.empty_like(d2)
        self.m.merge_device(gi.contiguous(), g1.contiguous(), g2.contiguous(), o_idx, o_d1, o_d2, init=init)
        self.m.synchronize()
        return o_idx, o_d1, o_d2
