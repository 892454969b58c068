"""Seeded synthetic frames (liborbx's orbx_synth_frame): identical bytes on every machine."""
import numpy as np

from ._lib import check, lib, np_ptr

SCENE, ZERO, CONST128, NOISE = 0, 1, 2, 3


def synth_frame(index, width=752, height=480, kind=SCENE, nrect=None, ncheck=None, right_view=False):
    """Frame `index` as a (height, width) uint8 array. Object counts default to 300 rectangles and
    40 checker patches per 752x480 of area."""
    area = (width * height) / float(752 * 480)
    if nrect is None:
        nrect = int(round(300 * area))
    if ncheck is None:
        ncheck = int(round(40 * area))
    out = np.empty((height, width), dtype=np.uint8)
    check(lib().orbx_synth_frame(int(index), width, height, int(kind), int(nrect), int(ncheck),
                                 1 if right_view else 0, np_ptr(out), out.strides[0]), "orbx_synth_frame")
    return out


def synth_batch(start, count, width=752, height=480, right_view=False):
    out = np.empty((count, height, width), dtype=np.uint8)
    for i in range(count):
        out[i] = synth_frame(start + i, width, height, right_view=right_view)
    return out


# ---- descriptor sets for the brute-force kNN workload (made on the device with torch ops) ----------
# Counter-based splitmix64, so any database row can be made from its index alone: row i is the four 64-bit
# outputs for counters 4i..4i+3 of the stream seeded 1. The first half of the queries are planted: query q copies
# database row src(q) and flips bits at r(q) in [0, 40] pseudo-random positions; the second half are uniform.
_GOLD, _M1, _M2 = 0x9E3779B97F4A7C15, 0xBF58476D1CE4E5B9, 0x94D049BB133111EB


def _s64(v):
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def splitmix64(counter, seed):
    """Output number `counter` (int64 torch tensor, taken as unsigned) of the splitmix64 stream seeded `seed`."""
    def lsr(x, k):
        return (x >> k) & ((1 << (64 - k)) - 1)
    z = (counter + 1) * _s64(_GOLD) + _s64(seed)
    z = (z ^ lsr(z, 30)) * _s64(_M1)
    z = (z ^ lsr(z, 27)) * _s64(_M2)
    return z ^ lsr(z, 31)


def knn_db_rows(first, count, device, piece=1 << 22):
    """Database rows first..first+count-1 as a (count, 32) uint8 tensor on `device`."""
    import torch
    out = torch.empty((count, 32), dtype=torch.uint8, device=device)
    for p0 in range(0, count, piece):
        n = min(piece, count - p0)
        c = torch.arange((first + p0) * 4, (first + p0 + n) * 4, dtype=torch.int64, device=device)
        out[p0:p0 + n] = splitmix64(c, 1).view(torch.uint8).view(n, 32)
    return out


def knn_rows_by_index(index):
    import torch
    c = (index[:, None] * 4 + torch.arange(4, dtype=torch.int64, device=index.device)[None, :]).reshape(-1)
    return splitmix64(c, 1).view(torch.uint8).view(index.shape[0], 32)


def knn_queries(nq, ndb, device):
    """-> queries (nq, 32) uint8, src (nq // 2,) int64 = planted database row, planted distance (nq // 2,) int32."""
    import torch
    half = nq // 2
    qid = torch.arange(half, dtype=torch.int64, device=device)
    src = (splitmix64(qid, 2) & ((1 << 62) - 1)) % ndb
    r = (splitmix64(qid, 3) & 0xFFFF) % 41
    mask = torch.zeros((half, 32), dtype=torch.uint8, device=device)
    flat = mask.view(-1)
    for k in range(40):
        bit = splitmix64(qid * 64 + k, 4) & 255
        on = k < r
        pos = (qid * 32 + (bit >> 3))[on]
        val = (torch.ones_like(bit) << (bit & 7)).to(torch.uint8)[on]
        flat[pos] = flat[pos] ^ val                   # one write per query per pass: no index collides
    lut = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int32, device=device)
    planted = lut[mask.to(torch.int64)].sum(dim=1).to(torch.int32)
    q = torch.empty((nq, 32), dtype=torch.uint8, device=device)
    q[:half] = knn_rows_by_index(src) ^ mask
    c = torch.arange((nq - half) * 4, dtype=torch.int64, device=device)
    q[half:] = splitmix64(c, 5).view(torch.uint8).view(nq - half, 32)
    return q, src, planted


def knn_top2_torch(q_row, db, base, piece=1 << 22):
    """Best two keys (distance << 40 | global row) of one query over a database shard with plain torch ops."""
    import torch
    lut = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int32, device=db.device)
    best = []
    for p0 in range(0, db.shape[0], piece):
        part = db[p0:p0 + piece]
        d = lut[(part ^ q_row[None, :]).to(torch.int64)].sum(dim=1)
        key = (d.to(torch.int64) << 40) | (torch.arange(part.shape[0], dtype=torch.int64, device=db.device) + base + p0)
        best.extend(torch.topk(key, min(2, key.shape[0]), largest=False).values.tolist())
    return sorted(best)[:2]
