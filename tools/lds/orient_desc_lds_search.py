#!/usr/bin/env python3
"""Searches the free choices of orient_desc_kernel's blur passes for few LDS bank conflicts:
  * the storage order of T's ten column groups (cells of one group stay consecutive: the column pass reads four in a row),
  * which work item sits on which lane of which round, in the row pass and in the column pass (items are independent,
    so any assignment computes the same patch; spare slots repeat an item of their own lane group, which costs nothing).
Cost model as tools/lds/orient_desc_lds_model.py (gfx950 lane groups and bank functions), reduced to multiplicities:
  row pass:    6 reads of R  -> per half-wave, distinct (24 j + gg - 1) mod 32;  one 16-byte store of T -> per 8 lanes, distinct cell mod 8
  column pass: 4 16-byte reads of T -> per 16-lane hardware group, distinct cell mod 16;  2 stores of Bl -> per half-wave, distinct (22 p + gg) mod 32
Prints the best order and the two item tables as C initialisers.

    python tools/lds/orient_desc_lds_search.py [seconds] [seed] [random orders besides the current one]
"""
import random
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__file__))
import orient_desc_lds_model as M

G16 = M.G128R                                   # lanes of the four 16-byte-read groups of one round
G8 = M.G128W
HALF = M.G32


def group_index(groups):
    idx = [0] * 64
    for g, lanes in enumerate(groups):
        for l in lanes:
            idx[l] = g
    return idx


IDX16, IDX8, IDXH = group_index(G16), group_index(G8), group_index(HALF)


def maxmult(vals):
    c = {}
    m = 0
    for v in vals:
        if v is None:
            continue
        c[v] = c.get(v, 0) + 1
        m = max(m, c[v])
    return max(m, 1) - 1


def excess(vals):
    """Lanes beyond the first on each value: falls with every collision removed, where maxmult only sees the worst one;
    the annealing adds it (lightly weighted) to the real cost so that plateaus have a slope."""
    c = {}
    for v in vals:
        if v is not None:
            c[v] = c.get(v, 0) + 1
    return sum(n - 1 for n in c.values())


def pass_cost(slots, keyA, groupsA, wA, keyB, groupsB, wB):
    """slots: 192 entries (item index or None). Cost = wA * sum over rounds and groupsA of extra(keyA) + wB * ... keyB."""
    total = 0
    for q in range(3):
        s = slots[64 * q:64 * q + 64]
        for g in groupsA:
            total += wA * maxmult([None if s[l] is None else keyA[s[l]] for l in g])
        for g in groupsB:
            total += wB * maxmult([None if s[l] is None else keyB[s[l]] for l in g])
    return total


def anneal(nitems, keyA, groupsA, wA, keyB, groupsB, wB, seconds, rng):
    slots = list(range(nitems)) + [None] * (192 - nitems)
    rng.shuffle(slots)
    ia, ib = group_index(groupsA), group_index(groupsB)

    def local(q, g, groups, key, w):
        vals = [None if slots[64 * q + l] is None else key[slots[64 * q + l]] for l in groups[g]]
        return w * (8 * maxmult(vals) + excess(vals))

    def real():
        return pass_cost(slots, keyA, groupsA, wA, keyB, groupsB, wB)

    def around(pos):
        q, l = divmod(pos, 64)
        return local(q, ia[l], groupsA, keyA, wA) + local(q, ib[l], groupsB, keyB, wB)

    cost = sum(local(q, g, groupsA, keyA, wA) for q in range(3) for g in range(len(groupsA))) + \
        sum(local(q, g, groupsB, keyB, wB) for q in range(3) for g in range(len(groupsB)))
    best, best_slots, best_real = cost, slots[:], real()
    t_end = time.time() + seconds
    T = 24.0
    it = 0
    while time.time() < t_end and best_real > 0:
        it += 1
        if it % 2000 == 0:
            T = max(0.3, T * 0.985)
        a, b = rng.randrange(192), rng.randrange(192)
        if a == b or (slots[a] is None and slots[b] is None):
            continue
        qa, la = divmod(a, 64)
        qb, lb = divmod(b, 64)
        same = qa == qb and ia[la] == ia[lb] and ib[la] == ib[lb]
        if same:
            continue
        before = around(a) + around(b)
        shared = 0
        if qa == qb and ia[la] == ia[lb]:
            shared += local(qa, ia[la], groupsA, keyA, wA)
        if qa == qb and ib[la] == ib[lb]:
            shared += local(qa, ib[la], groupsB, keyB, wB)
        before -= shared
        slots[a], slots[b] = slots[b], slots[a]
        after = around(a) + around(b)
        shared2 = 0
        if qa == qb and ia[la] == ia[lb]:
            shared2 += local(qa, ia[la], groupsA, keyA, wA)
        if qa == qb and ib[la] == ib[lb]:
            shared2 += local(qa, ib[la], groupsB, keyB, wB)
        after -= shared2
        d = after - before
        if d <= 0 or rng.random() < pow(2.718281828, -d / T):
            cost += d
            if cost < best:
                best, best_slots, best_real = cost, slots[:], real()
        else:
            slots[a], slots[b] = slots[b], slots[a]
    return best_real, best_slots


def fill_spares(slots, groups):
    """A spare slot repeats an item of its own lane group (the narrowest one of its pass; that group lies inside one
    half-wave) of the same round: same addresses as another lane of the group, no extra LDS cycle."""
    out = slots[:]
    idx = group_index(groups)
    for pos, s in enumerate(out):
        if s is not None:
            continue
        q, l = divmod(pos, 64)
        for cand in list(groups[idx[l]]) + list(HALF[IDXH[l]]):
            if slots[64 * q + cand] is not None:
                out[pos] = slots[64 * q + cand]
                break
        else:
            out[pos] = 0
    return out


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    orders = [(2, 4, 3, 5, 10, 1, 8, 6, 7, 9)]
    for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 11):
        o = list(range(1, 11))
        rng.shuffle(o)
        orders.append(tuple(o))
    best = None
    per = seconds / len(orders) / 2
    for order in orders:
        h, v, cells = M.items(order)
        # row pass: item = (R offset, T cell offset)
        rkeyA = [(a // 4) % 32 for a, _ in h]                      # 24 j + gg - 1 mod 32 (dword of the first read)
        rkeyB = [(t // 16) % 8 for _, t in h]
        rc, rslots = anneal(len(h), rkeyA, HALF, 6, rkeyB, G8, 1, per, rng)
        ckeyA = [(t // 16) % 16 for t, _ in v]
        ckeyB = [(b // 4) % 32 for _, b in v]
        cc, cslots = anneal(len(v), ckeyA, G16, 4, ckeyB, HALF, 2, per, rng)
        print("order", order, "row pass", rc, "column pass", cc, "total", rc + cc, flush=True)
        if best is None or rc + cc < best[0]:
            best = (rc + cc, order, fill_spares(rslots, G8), fill_spares(cslots, G16), h, v)
    total, order, rslots, cslots, h, v = best
    print("\nbest total extra LDS cycles per keypoint (blur passes):", total)
    print("constexpr int kTGroupOrder[10] = {%s};" % ", ".join(str(g) for g in order))
    print("// item index (in the natural order: row pair / blurred row pair ascending, then column group) per slot")
    print("constexpr unsigned char kRowItemSlot[192] = {%s};" % ", ".join(str(s) for s in rslots))
    print("constexpr unsigned char kColItemSlot[192] = {%s};" % ", ".join(str(s) for s in cslots))
    # cross-check with the address-level model
    hs = [h[s] for s in rslots]
    vs = [v[s] for s in cslots]
    chk = {"row_reads": 0, "row_writes": 0, "col_reads": 0, "col_writes": 0}
    for q in range(3):
        it = hs[64 * q:64 * q + 64]
        for hh in range(2):
            for k in range(3):
                chk["row_reads"] += M.extra_cycles([a + 48 * hh + 4 * k for a, _ in it], M.G32, 1, 32)
        chk["row_writes"] += M.extra_cycles([M.kTOffset + t for _, t in it], M.G128W, 4, 32)
        it = vs[64 * q:64 * q + 64]
        for r in range(4):
            chk["col_reads"] += M.extra_cycles([M.kTOffset + t + 16 * r for t, _ in it], M.G128R, 4, 64)
        for k in range(2):
            chk["col_writes"] += M.extra_cycles([b + M.kTPitch * k for _, b in it], M.G32, 1, 32)
    print("address-level check:", chk, sum(chk.values()))


if __name__ == "__main__":
    main()
