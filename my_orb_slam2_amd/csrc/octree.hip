// Quadtree keypoint distribution (ORBextractor::DistributeOctTree and ExtractorNode::DivideNode,
// src/ORBextractor.cc:473-755), one 256-thread workgroup per (frame, level).
//
// The reference keeps a std::list of nodes, each owning a vector of keypoints, and splits nodes
// one at a time. Here candidates never move: each carries the index of the node that owns it, and
// a node's position in the array IS its position in the reference's list. One "round" is either a
// full sweep of the reference's outer loop (:598-657) or one pass of its size-ordered loop
// (:668-729); both have the same shape:
//   - every node with more than one point is a split candidate;
//   - they are processed in list order (sweep) or by point count descending, ties in list order
//     (the reference sorts (count, node address) ascending and walks backwards, :676-677; the
//     address is replaced by creation order, and since new nodes are pushed to the front, "newest
//     first" is list order);
//   - processing stops once the list holds N nodes (only in the size-ordered loop, :722-723);
//   - each processed node is replaced by its non-empty children n1..n4, pushed to the front of the
//     list one by one (:612-652), so the new list is: all children in reverse creation order,
//     then the surviving old nodes in their old order.
// Child point counts of every candidate are counted speculatively (LDS atomics) by the same pass that
// gives the points their node, so they are ready when a round starts; prefix sums give the stopping
// point and every node's new position, and one pass over the points per round re-labels them and
// counts for the round after. Finally each node emits its strongest point, first one on ties (:736-752).
#include "orbx_internal.h"

namespace orbx {

namespace {

constexpr int kThreadsBatch = 256;   // workgroup size in batched calls: many (frame, level) workgroups share the chip
constexpr int kThreadsFew = 1024;    // calls of a few frames: the chip is nearly empty and a level's passes over its
                                     // candidates are dependent memory round trips, so more threads = fewer trips each
constexpr int kBatch = 4;          // candidates a thread loads before it uses the first (hides the memory latency)

// Inclusive scan over the 64 lanes of a wave with DPP adds (no LDS traffic): shifts by 1, 2, 4, 8 inside each row of
// 16 lanes, then the totals of rows 0 and 2 are broadcast into rows 1 and 3, and the total of the first half into
// the second.
__device__ __forceinline__ int wave_scan_inclusive(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);  // row_bcast31 into rows 2 and 3
    return x;
}

// Inclusive scan of data[0..n) in place. `part` holds kThreads ints. All threads must call.
// Each thread owns a run of ceil(n / 256) elements (one element in all but the largest quotas); run totals are
// scanned inside each wave and the four wave totals are combined through LDS: two barriers in all.
template <int kThreads>
__device__ void block_scan_inclusive(int* data, int n, int* part) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int per = (n + kThreads - 1) / kThreads;
    const int b = min(t * per, n), e = min(b + per, n);
    int s = 0;
    for (int i = b; i < e; i++) s += data[i];
    const int incl = wave_scan_inclusive(s);
    if (lane == 63) part[wv] = incl;
    __syncthreads();
    int run = incl - s;
    for (int w = 0; w < wv; w++) run += part[w];
    for (int i = b; i < e; i++) {
        run += data[i];
        data[i] = run;
    }
    __syncthreads();
}

// The scan a sweep round needs, with its input made on the fly: node i contributes 1 | (children - 1) << 16 when it
// holds more than one point (it is split; cc holds its four child counts) and 0 otherwise. data[i] receives the
// inclusive sums: low half = split nodes up to and including i, high half = the (children - 1) of those. Both stay
// below 2^16 (the node table has fewer than 2^14 entries). Two barriers; *zeroed is cleared between them.
template <int kThreads>
__device__ void sweep_scan(const int* cnt, const int* cc, int* data, int n, int* part, int* zeroed) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int per = (n + kThreads - 1) / kThreads;
    const int b = min(t * per, n), e = min(b + per, n);
    int s = 0;
    for (int i = b; i < e; i++) {
        int v = 0;
        if (cnt[i] > 1) {
            const int* c4 = cc + 4 * i;
            v = 1 | (((c4[0] > 0) + (c4[1] > 0) + (c4[2] > 0) + (c4[3] > 0) - 1) << 16);
        }
        data[i] = v;
        s += v;
    }
    const int incl = wave_scan_inclusive(s);
    if (lane == 63) part[wv] = incl;
    __syncthreads();
    if (t == 0) *zeroed = 0;              // (every thread has read the old value before it reached the barrier above)
    int run = incl - s;
    for (int w = 0; w < wv; w++) run += part[w];
    for (int i = b; i < e; i++) {
        run += data[i];
        data[i] = run;
    }
    __syncthreads();
}

// A node's box is one 8-byte LDS word: x = x0 | y0 << 16, y = x1 | y1 << 16 (coordinates are below 2^16).
struct Nodes {
    uint2* box;
    int* cnt;
};

__device__ __forceinline__ uint2 make_box(int x0, int y0, int x1, int y1) {
    return make_uint2((uint32_t)x0 | ((uint32_t)y0 << 16), (uint32_t)x1 | ((uint32_t)y1 << 16));
}

__device__ __forceinline__ int quadrant(uint32_t c, const Nodes& nd, int n) {
    const uint2 b = nd.box[n];
    const int x0 = (int)(b.x & 0xFFFFu), y0 = (int)(b.x >> 16), x1 = (int)(b.y & 0xFFFFu), y1 = (int)(b.y >> 16);
    const int xm = x0 + ((x1 - x0 + 1) >> 1);  // UL.x + ceil((UR.x-UL.x)/2) (:475)
    const int ym = y0 + ((y1 - y0 + 1) >> 1);
    return (cand_x(c) < xm ? 0 : 1) + (cand_y(c) < ym ? 0 : 2);   // n1, n2, n3, n4 (:507-517)
}

// The whole selection for one level, with the level's candidates at `cand` and their node indices at `nid`.
template <int kThreads>
__device__ void distribute(uint32_t* cand, uint16_t* nid, int P, const LevelGeom& L, int nodeCap,
                           uint8_t* lds, uint32_t* selOut, int32_t* selCountOut) {
    const int t = threadIdx.x;
    const int N = L.quota;
    // ---- carve LDS ----
    Nodes A, B;
    short* sp = reinterpret_cast<short*>(lds);
    A.box = reinterpret_cast<uint2*>(lds);
    B.box = A.box + nodeCap;
    short* childPos = sp + 8 * nodeCap;              // [node][4]
    int* ip = reinterpret_cast<int*>(sp + 12 * nodeCap);
    A.cnt = ip; B.cnt = ip + nodeCap;
    int* cc = ip + 2 * nodeCap;                      // [node][4] speculative child counts of the current nodes
    int* rank = ip + 6 * nodeCap;                    // processing rank of a split candidate
    int* order = ip + 7 * nodeCap;                   // node at processing rank r
    int* sinc = ip + 8 * nodeCap;                    // inclusive sum of (children - 1) in processing order
    int* keptPos = ip + 9 * nodeCap;                 // scan of surviving old nodes / new position
    int* ccNext = ip + 10 * nodeCap;                 // [node][4] the same counts for the nodes of the next round
    int* part = ip + 14 * nodeCap;                   // kThreads ints of scan scratch
    int* sh = part + kThreads;                       // small shared scalars
    // sh[0]=nsplit sh[1]=m sh[2]=nExp

    // ---- root nodes (:544-577) ----
    const int nIni = L.nIni;
    const int height = (L.h - kBorder) - kBorder;
    for (int i = t; i < nIni; i += kThreads) {
        B.box[i] = make_box((int)(L.hX * (float)i), 0, (int)(L.hX * (float)(i + 1)), height);
        B.cnt[i] = 0;
    }
    __syncthreads();
    for (int p0 = t; p0 < P; p0 += kThreads * kBatch) {
        uint32_t c[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; k++) c[k] = cand[min(p0 + k * kThreads, P - 1)];
#pragma unroll
        for (int k = 0; k < kBatch; k++) {
            if (p0 + k * kThreads >= P) break;
            int r = (int)((float)cand_x(c[k]) / L.hX);
            r = min(max(r, 0), nIni - 1);
            nid[p0 + k * kThreads] = (uint16_t)r;
            atomicAdd(&B.cnt[r], 1);
        }
    }
    __syncthreads();
    // drop empty roots, keep order
    for (int i = t; i < nIni; i += kThreads) keptPos[i] = B.cnt[i] > 0 ? 1 : 0;
    __syncthreads();
    block_scan_inclusive<kThreads>(keptPos, nIni, part);
    int Lc = nIni ? keptPos[nIni - 1] : 0;
    for (int i = t; i < nIni; i += kThreads) {
        if (B.cnt[i] > 0) {
            const int pos = keptPos[i] - 1;
            A.box[pos] = B.box[i];
            A.cnt[pos] = B.cnt[i];
            cc[4 * pos] = cc[4 * pos + 1] = cc[4 * pos + 2] = cc[4 * pos + 3] = 0;
        }
    }
    __syncthreads();
    // Every pass that gives the points their node also counts, for nodes with more than one point, how many
    // fall into each quadrant: the child counts a round needs are ready when it starts.
    for (int p0 = t; p0 < P; p0 += kThreads * kBatch) {
        int r[kBatch];
        uint32_t c[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; k++) {
            const int p = min(p0 + k * kThreads, P - 1);
            r[k] = nid[p];
            c[k] = cand[p];
        }
#pragma unroll
        for (int k = 0; k < kBatch; k++) {
            if (p0 + k * kThreads >= P) break;
            const int n = keptPos[r[k]] - 1;
            nid[p0 + k * kThreads] = (uint16_t)n;
            if (A.cnt[n] > 1) atomicAdd(&cc[4 * n + quadrant(c[k], A, n)], 1);
        }
    }
    __syncthreads();

    Nodes cur = A, nxt = B;
    bool careful = false;
    for (int round = 0; round < 4096; round++) {
        int C, Lnew;
        bool done;
        // Writes the children of split node i (first child = creation index k) and the table entry of a node that
        // stays (at position pos) into nxt, and records where the points of node i go.
        auto emit_split = [&](int i, int k) {
            const uint2 b = cur.box[i];
            const int x0 = (int)(b.x & 0xFFFFu), y0 = (int)(b.x >> 16), x1 = (int)(b.y & 0xFFFFu), y1 = (int)(b.y >> 16);
            const int xm = x0 + ((x1 - x0 + 1) >> 1), ym = y0 + ((y1 - y0 + 1) >> 1);
            int nExp = 0;
            for (int q = 0; q < 4; q++) {
                const int cq = cc[4 * i + q];
                if (cq == 0) continue;
                const int pos = C - 1 - k;           // each child is pushed to the front
                k++;
                nxt.box[pos] = make_box((q & 1) ? xm : x0, (q & 2) ? ym : y0, (q & 1) ? x1 : xm, (q & 2) ? y1 : ym);
                nxt.cnt[pos] = cq;
                ccNext[4 * pos] = ccNext[4 * pos + 1] = ccNext[4 * pos + 2] = ccNext[4 * pos + 3] = 0;
                childPos[4 * i + q] = (short)pos;
                nExp += cq > 1;
            }
            keptPos[i] = -1;
            if (nExp) atomicAdd(&sh[2], nExp);
        };
        auto emit_kept = [&](int i, int pos) {
            nxt.box[pos] = cur.box[i];
            nxt.cnt[pos] = cur.cnt[i];
            ccNext[4 * pos] = ccNext[4 * pos + 1] = ccNext[4 * pos + 2] = ccNext[4 * pos + 3] = 0;
            keptPos[i] = pos;
        };
        if (!careful) {
            // ---- a sweep: every node with more than one point is split, in list order, so one scan over the nodes
            // gives every split node its rank and the creation index of its first child, and every other node the
            // number of unsplit nodes before it ----
            sweep_scan<kThreads>(cur.cnt, cc, rank, Lc, part, &sh[2]);
            const int tot = Lc ? rank[Lc - 1] : 0;
            const int m = tot & 0xFFFF;
            C = (tot >> 16) + m;                         // children created this round
            Lnew = Lc - m + C;
            done = Lnew >= N || Lnew == Lc;              // (:661)
            if (Lnew > nodeCap) {  // cannot happen (Lnew <= max(N+3, 4*nIni)); never write out of bounds
                if (t == 0) *selCountOut = 0;
                return;
            }
            for (int i = t; i < Lc; i += kThreads) {
                const int S = rank[i], before = S & 0xFFFF;
                if (cur.cnt[i] > 1) {
                    const int* c4 = cc + 4 * i;
                    const int extra = (c4[0] > 0) + (c4[1] > 0) + (c4[2] > 0) + (c4[3] > 0) - 1;
                    emit_split(i, (S >> 16) - extra + before - 1);
                } else {
                    emit_kept(i, C + i - before);
                }
            }
            __syncthreads();
        } else {
        // ---- processing order: by point count, ties in list order ----
        if (P >= (1 << 20)) {
            // (a node could hold more points than the packed keys below have bits for: compare count and position)
            if (t == 0) sh[1] = 0;
            __syncthreads();
            for (int i = t; i < Lc; i += kThreads) {
                const int ci = cur.cnt[i];
                if (ci > 1) {
                    int r = 0;
                    for (int j = 0; j < Lc; j++) {
                        const int cj = cur.cnt[j];
                        r += (cj > 1 && (cj > ci || (cj == ci && j < i))) ? 1 : 0;
                    }
                    rank[i] = r;
                    order[r] = i;
                    atomicAdd(&sh[1], 1);
                }
            }
        } else {
            // Rank = number of split candidates with a larger key, key = count << 12 | (4095 - list position): more
            // points first, list order among equals. Keys sit in LDS (sinc is free until the ranks are known), padded
            // with zeros to a multiple of four so that they are read four at a time, every lane the same address.
            unsigned* key = reinterpret_cast<unsigned*>(sinc);
            const int Lc4 = (Lc + 3) & ~3;
            if (t == 0) sh[1] = 0;
            for (int i = t; i < Lc4; i += kThreads) {
                const int ci = i < Lc ? cur.cnt[i] : 0;
                key[i] = ci > 1 ? ((unsigned)ci << 12) | (unsigned)(4095 - i) : 0u;
            }
            __syncthreads();
            for (int i0 = 0; i0 < Lc; i0 += kThreads) {
                const int i = i0 + t;
                const unsigned ki = i < Lc ? key[i] : 0u;
                int r = 0;
                for (int j = 0; j < Lc4; j += 4) {
                    const uint4 k4 = *reinterpret_cast<const uint4*>(key + j);
                    r += (k4.x > ki) + (k4.y > ki) + (k4.z > ki) + (k4.w > ki);
                }
                if (ki != 0u) {
                    rank[i] = r;
                    order[r] = i;
                    atomicAdd(&sh[1], 1);
                }
            }
        }
        __syncthreads();
        const int m = sh[1];
        // ---- how many candidates are actually split ----
        for (int r = t; r < m; r += kThreads) {
            const int* c4 = cc + 4 * order[r];
            sinc[r] = (c4[0] > 0) + (c4[1] > 0) + (c4[2] > 0) + (c4[3] > 0) - 1;
        }
        __syncthreads();
        block_scan_inclusive<kThreads>(sinc, m, part);
        if (t == 0) sh[0] = m;
        __syncthreads();
        for (int r = t; r < m; r += kThreads)
            if (Lc + sinc[r] >= N && (r == 0 || Lc + sinc[r - 1] < N)) sh[0] = r + 1;  // first rank reaching N (:722)
        __syncthreads();
        const int nsplit = sh[0];
        C = nsplit > 0 ? sinc[nsplit - 1] + nsplit : 0;   // children created this round
        Lnew = Lc - nsplit + C;
        done = Lnew >= N || Lnew == Lc;     // (:726)
        if (Lnew > nodeCap) {  // cannot happen (Lnew <= max(N+3, 4*nIni)); never write out of bounds
            if (t == 0) *selCountOut = 0;
            return;
        }
        // ---- new node table ----
        for (int i = t; i < Lc; i += kThreads) keptPos[i] = (cur.cnt[i] > 1 && rank[i] < nsplit) ? 0 : 1;
        __syncthreads();
        block_scan_inclusive<kThreads>(keptPos, Lc, part);
        if (t == 0) sh[2] = 0;
        __syncthreads();
        for (int i = t; i < Lc; i += kThreads) {
            if (cur.cnt[i] > 1 && rank[i] < nsplit) {
                const int r = rank[i];
                emit_split(i, r > 0 ? sinc[r - 1] + r : 0);
            } else {
                emit_kept(i, C + keptPos[i] - 1);
            }
        }
        __syncthreads();
        }
        // ---- re-label the points (and count them into the quadrants of their new nodes for the next round) ----
        for (int p0 = t; p0 < P; p0 += kThreads * kBatch) {
            int n[kBatch];
            uint32_t c[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; k++) {
                const int p = min(p0 + k * kThreads, P - 1);
                n[k] = nid[p];
                c[k] = cand[p];
            }
#pragma unroll
            for (int k = 0; k < kBatch; k++) {
                if (p0 + k * kThreads >= P) break;
                const int kp = keptPos[n[k]];
                const int nn = kp >= 0 ? kp : childPos[4 * n[k] + quadrant(c[k], cur, n[k])];
                nid[p0 + k * kThreads] = (uint16_t)nn;
                if (!done && nxt.cnt[nn] > 1) atomicAdd(&ccNext[4 * nn + quadrant(c[k], nxt, nn)], 1);
            }
        }
        __syncthreads();
        const int nExp = sh[2];
        const Nodes tmp = cur; cur = nxt; nxt = tmp;
        int* const tc = cc; cc = ccNext; ccNext = tc;
        if (!careful && !done && Lnew + 3 * nExp > N) careful = true;   // (:665)
        Lc = Lnew;
        if (done) break;
    }

    // ---- strongest point of every node, first one on ties (:736-752) ----
    int* best = cc;
    for (int i = t; i < Lc; i += kThreads) best[i] = 0;
    __syncthreads();
    for (int p0 = t; p0 < P; p0 += kThreads * kBatch) {
        int n[kBatch];
        uint32_t c[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; k++) {
            const int p = min(p0 + k * kThreads, P - 1);
            n[k] = nid[p];
            c[k] = cand[p];
        }
#pragma unroll
        for (int k = 0; k < kBatch; k++)
            if (p0 + k * kThreads < P)
                atomicMax(reinterpret_cast<unsigned*>(&best[n[k]]),
                          ((unsigned)cand_score(c[k]) << 24) | (0xFFFFFFu - (unsigned)(p0 + k * kThreads)));
    }
    __syncthreads();
    for (int i = t; i < Lc; i += kThreads) {
        const unsigned p = 0xFFFFFFu - ((unsigned)best[i] & 0xFFFFFFu);
        if (i < L.selCap) selOut[i] = cand[p];
    }
    if (t == 0) *selCountOut = min(Lc, L.selCap);
}

}  // namespace

namespace {
size_t lds_bytes(const PyramidGeom& g, int threads) {
    // 12 shorts + 14 ints per node, scan scratch, scalars
    return (size_t)g.nodeCap * (12 * 2 + 14 * 4) + (threads + 8) * 4 + 16;
}
}  // namespace

size_t octree_lds_bytes(const PyramidGeom& g) { return lds_bytes(g, kThreadsFew); }

template <int kThreads>
__global__ __launch_bounds__(kThreads) void octree_kernel(FrameSet fs, PyramidGeom g) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // This kernel waits far more than it computes; when its waves share a SIMD with the compute-bound kernels of
    // other streams they go first whenever they are ready (a small, repeatable gain in the overlapped arrangement;
    // the same setting in the resize kernel made that kernel slower).
    __builtin_amdgcn_s_setprio(3);
    const int t = threadIdx.x;
    const int f = blockIdx.x, l = blockIdx.y;
    const LevelGeom& L = g.lv[l];
    uint32_t* selOut = fs.sel + (long long)f * g.totalSel + L.selBase;
    int32_t* selCountOut = fs.selCount + (long long)f * g.nlevels + l;
    const int nCells = L.nCols * L.nRows;
    if (nCells == 0) {   // (a quota of 0 still keeps the children of the first sweep, as in the reference)
        if (t == 0) *selCountOut = 0;
        return;
    }
    // ---- gather the level's candidates in the reference's order: cells row-major (:781-821) ----
    int* part = reinterpret_cast<int*>(smem);            // scan scratch, reused by distribute()
    const int32_t* counts = fs.cellCount + (long long)f * g.totalCells + L.cellBase;
    const uint32_t* slots = fs.slots + (long long)f * g.totalSlots + L.slotBase;
    const int per = (nCells + kThreads - 1) / kThreads;
    const int cb = min(t * per, nCells), ce = min(cb + per, nCells);
    int mine = 0;
    for (int c = cb; c < ce; c++) mine += counts[c];
    int incl = mine;
    {
        const int lane = t & 63, wv = t >> 6;
        incl = wave_scan_inclusive(incl);
        if (lane == 63) part[wv] = incl;
        __syncthreads();
        for (int w = 0; w < wv; w++) incl += part[w];
    }
    int P = 0;
    for (int w = 0; w < kThreads / 64; w++) P += part[w];
    int dst = incl - mine;
    __syncthreads();
    if (P == 0) {
        if (t == 0) *selCountOut = 0;
        return;
    }
    // The gathered list and the node index of every candidate stay in global memory (they are L2 hits):
    // keeping them in LDS was no faster, and the smaller LDS footprint lets workgroups of the kernels queued
    // on other streams share the CU while this latency-bound kernel waits.
    uint32_t* cand = fs.cand + (long long)f * g.totalSlots + L.slotBase;
    uint16_t* nid = fs.nid + (long long)f * g.totalSlots + L.slotBase;
    // A thread's first two cells (all it has on most levels) go together: their first eight entries are requested
    // before either is stored.
    int cFrom = cb;
    if (ce - cb >= 2) {
        const int n0 = counts[cb], n1 = counts[cb + 1];
        const uint32_t* s0 = slots + (long long)cb * L.slotCap;
        const uint32_t* s1 = s0 + L.slotCap;
        uint32_t v0[8], v1[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            v0[k] = k < n0 ? s0[k] : 0u;
            v1[k] = k < n1 ? s1[k] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k < n0) cand[dst + k] = v0[k];
        for (int k = 8; k < n0; k++) cand[dst + k] = s0[k];
        dst += n0;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k < n1) cand[dst + k] = v1[k];
        for (int k = 8; k < n1; k++) cand[dst + k] = s1[k];
        dst += n1;
        cFrom = cb + 2;
    }
    for (int c = cFrom; c < ce; c++) {
        const int n = counts[c];
        const uint32_t* src = slots + (long long)c * L.slotCap;
        // eight loads in flight per batch instead of one dependent load per entry
        for (int k0 = 0; k0 < n; k0 += 8) {
            uint32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = k0 + k < n ? src[k0 + k] : 0u;
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (k0 + k < n) cand[dst + k] = v[k];
            dst += min(8, n - k0);
        }
    }
    __syncthreads();
    distribute<kThreads>(cand, nid, P, L, g.nodeCap, smem, selOut, selCountOut);
}

hipError_t launch_octree(const FrameSet& fs, const PyramidGeom& g, int nframes, hipStream_t s) {
    const bool few = nframes <= 8;
    const int threads = few ? kThreadsFew : kThreadsBatch;
    dim3 grid(nframes, g.nlevels, 1), block(threads, 1, 1);
    const size_t lds = lds_bytes(g, threads);
    auto launch = [&](auto kernel) -> hipError_t {
        if (lds > 64 * 1024) {   // above the default dynamic-LDS limit: raise it for this kernel
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kernel, grid, block, lds, s, fs, g);
        return hipGetLastError();
    };
    return few ? launch(octree_kernel<kThreadsFew>) : launch(octree_kernel<kThreadsBatch>);
}

}  // namespace orbx
