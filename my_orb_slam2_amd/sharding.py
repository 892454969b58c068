"""One process per GPU: how frames, stereo pairs and a large descriptor database are split.

Extraction and stereo matching need no communication (frame / pair k belongs to rank k mod G).
The brute-force kNN over a sharded database has exactly one exchange step: every rank scans its
contiguous shard of database rows for all queries into one packed 16-byte record per query
({int32 d1, d2; int64 idx}, orbx_knn2_hit), ONE all-gather moves every rank's nq records (RCCL when the
process group is "nccl"), and every rank folds them with orbx_knn2_merge_packed_device.
The fold is exact: the two smallest distances of the whole database lie inside the union of the
per-shard best two, and shards are folded in row order so the lowest row index wins ties, as in
the sequential scan of my_orb_slam2/src/ORBmatcher.cc:204-229.
Scan, collective and fold are ordered on the device by events; the host never waits in between.
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


def pack_hits(idx, d1, d2):
    """(nq,) int64 / int32 / int32 -> (nq, 2) int64 records with the layout of orbx_knn2_hit
    (little endian: d1 in the low half of word 0, d2 in the high half, idx in word 1)."""
    w0 = (d1.to(torch.int64) & 0xFFFFFFFF) | (d2.to(torch.int64) << 32)
    return torch.stack((w0, idx.to(torch.int64)), dim=1).contiguous()


def unpack_hits(hits):
    """Inverse of pack_hits for (..., 2) int64 records -> idx, d1, d2."""
    w0 = hits[..., 0]
    d1 = ((w0 << 32) >> 32).to(torch.int32)
    d2 = (w0 >> 32).to(torch.int32)
    return hits[..., 1].contiguous(), d1, d2


def all_gather_hits(hits, group=None):
    """One all-gather of every rank's (nq, 2) int64 records -> (world, nq, 2), shard order = rank order.
    With a backend that cannot move device memory (gloo rehearsals) the records are staged through the host."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return hits[None]
    backend = dist.get_backend(group)
    src = hits.contiguous()
    if src.is_cuda and backend != "nccl":
        src = src.cpu()
    out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src, group=group)            # rank r's records at rows [r * nq, (r + 1) * nq)
    return out.view((world,) + tuple(src.shape)).to(hits.device)


class ShardedHammingMatcher:
    """kNN top-2 against a database whose rows are split across the ranks of a process group."""

    def __init__(self, d_db_shard, index_base, device=0, group=None):
        """d_db_shard: this rank's (rows, 32) uint8 CUDA tensor; index_base: global index of its row 0."""
        self.m = HammingMatcher(device)
        self.db = d_db_shard
        self.base = int(index_base)
        self.group = group

    def knn2(self, d_query, init=256, slice_rows=None):
        """-> (idx, d1, d2) of the scan over the whole database, on every rank. Returns without waiting for the
        device; the results are ordered behind torch's current stream. slice_rows: queries per kernel launch
        (None = all at once)."""
        nq = d_query.shape[0]
        dev = d_query.device
        hits = torch.empty((nq, 2), dtype=torch.int64, device=dev)
        step = nq if not slice_rows else int(slice_rows)
        for s0 in range(0, nq, step):
            s1 = min(nq, s0 + step)
            self.m.knn2_packed_device(d_query[s0:s1], self.db, hits[s0:s1], init=init, index_base=self.base)
        self.m.after()                                   # the collective starts after the scan (device-side order)
        gathered = all_gather_hits(hits, self.group)     # ONE collective of nq * 16 bytes per rank
        world = gathered.shape[0]
        o_idx = torch.empty(nq, dtype=torch.int64, device=dev)
        o_d1 = torch.empty(nq, dtype=torch.int32, device=dev)
        o_d2 = torch.empty(nq, dtype=torch.int32, device=dev)
        self.m.merge_packed_device(gathered, world, o_idx, o_d1, o_d2, init=init)
        self.m.after()
        return o_idx, o_d1, o_d2
