#!/usr/bin/env python3
"""Bank-conflict model of orient_desc_kernel's LDS traffic, with the lane groups and bank functions measured on
gfx950 (ds_read_b32: 2 groups of 32 lanes, 32 banks; ds_read_b128: 4 groups of 16 lanes in the order
{0-3,12-15,20-27}, {4-11,16-19,28-31}, ... and 64 banks; ds_write_b128: 8 groups of 8 contiguous lanes, 32 banks).
Rebuilds the kernel's compile-time tables (which (row pair, column group) cells of T exist, the work items of the two
blur passes), counts the extra LDS cycles of every access for a given storage order of T's column groups, and
searches the 10! orders for the one with the fewest.

    python tools/lds/orient_desc_lds_model.py            # report the current order and the best found
"""
import itertools
import sys

kCx, kBlRows, kTPairs, kRPitch, kTPitch = 23, 37, 22, 48, 44
kTOffset = 43 * kRPitch


def reachable(dx, dy):
    nx = -2 * dx - 1 if dx < 0 else 2 * dx - 1
    ny = -2 * dy - 1 if dy < 0 else 2 * dy - 1
    px, py = max(nx, 0), max(ny, 0)
    return px * px + py * py <= 4 * 338 + 2


def blur_group_needed(y, gg):
    if y < 0 or y >= kBlRows:
        return False
    return any(reachable(4 * gg + b - kCx, y - 18) for b in range(4))


def rowsum_group_needed(r, gg):
    return any(blur_group_needed(y, gg) for y in range(r - 6, r + 1))


def layout(order):
    lo, hi, start = {}, {}, {}
    for gg in range(1, 11):
        js = [j for j in range(kTPairs) if rowsum_group_needed(2 * j, gg) or rowsum_group_needed(2 * j + 1, gg)]
        lo[gg], hi[gg] = min(js), max(js)
    at = 0
    for gg in order:
        start[gg] = at
        at += hi[gg] - lo[gg] + 1
    return lo, hi, start, at


def items(order):
    lo, hi, start, cells = layout(order)
    cell = lambda j, gg: 16 * (start[gg] + j - lo[gg])
    h, v = [], []
    for j in range(kTPairs):
        for gg in range(1, 11):
            if rowsum_group_needed(2 * j, gg) or rowsum_group_needed(2 * j + 1, gg):
                h.append((2 * j * kRPitch + (gg - 1) * 4, cell(j, gg)))
    for p in range((kBlRows + 1) // 2):
        for gg in range(1, 11):
            if blur_group_needed(2 * p, gg) or blur_group_needed(2 * p + 1, gg):
                v.append((cell(p, gg), 2 * p * kTPitch + 4 * gg))
    return h, v, cells


G32 = [list(range(0, 32)), list(range(32, 64))]
G128R = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G128R += [[l + 32 for l in g] for g in G128R]
G128W = [list(range(8 * k, 8 * k + 8)) for k in range(8)]


def extra_cycles(addrs, groups, width_dwords, nbanks):
    """addrs: byte address per lane (None = inactive). Extra cycles = sum over groups of (max distinct dwords on a bank - 1)."""
    total = 0
    for g in groups:
        per_bank = {}
        for l in g:
            a = addrs[l]
            if a is None:
                continue
            for w in range(width_dwords):
                d = a // 4 + w
                per_bank.setdefault(d % nbanks, set()).add(d)
        if per_bank:
            total += max(len(s) for s in per_bank.values()) - 1
    return total


def model(order, verbose=False):
    h, v, cells = items(order)
    pad = lambda L, n: L + [L[0]] * (n - len(L))
    h, v = pad(h, 192), pad(v, 192)
    out = {"row_reads": 0, "row_writes": 0, "col_reads": 0, "col_writes": 0}
    for q in range(3):
        it = h[64 * q:64 * q + 64]
        for hh in range(2):
            for k in range(3):
                out["row_reads"] += extra_cycles([a + 48 * hh + 4 * k for a, _ in it], G32, 1, 32)
        out["row_writes"] += extra_cycles([kTOffset + t for _, t in it], G128W, 4, 32)
        it = v[64 * q:64 * q + 64]
        for r in range(4):
            out["col_reads"] += extra_cycles([kTOffset + t + 16 * r for t, _ in it], G128R, 4, 64)
        for k in range(2):
            out["col_writes"] += extra_cycles([b + kTPitch * k for _, b in it], G32, 1, 32)
    out["total"] = sum(out.values())
    return out


if __name__ == "__main__":
    cur = (2, 4, 3, 5, 10, 1, 8, 6, 7, 9)
    print("current order", cur, model(cur))
    best = None
    for n, order in enumerate(itertools.permutations(range(1, 11))):
        m = model(order)
        key = (m["row_writes"] + m["col_reads"], m["total"])
        if best is None or key < best[0]:
            best = (key, order, m)
            print("better", order, m, flush=True)
        if len(sys.argv) > 1 and n >= int(sys.argv[1]):
            break
    print("best", best[1], best[2])
