// ORBmatcher::SearchByBoW, both overloads (src/ORBmatcher.cc:153-293 KeyFrame x Frame, :532-690
// KeyFrame x KeyFrame): features of the two sides are compared only inside the same vocabulary node.
//
// The distances are the data-parallel part: for every common node and every left feature that owns a live
// map point, one wave computes the 256-bit distances to the node's right-hand features. What the reference
// does with them is a greedy pass whose result depends on the visiting order (a right-hand feature taken
// by an earlier match is skipped), so that part is replayed on the host over the distance rows, line by
// line as written there.
#include "orbx_internal.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace orbx {

// One wave per row: left feature rows[r].idx1 against items2[rows[r].start2 .. +count2). Besides the whole
// distance row the wave leaves a short list: the row's kBowTop smallest entries among the right-hand features
// that may be matched at all (valid2, when given), ordered by (distance, position), as distance << 16 | position,
// closed by kBowListEnd where the row has fewer. The host pass needs the rest of the row only when earlier
// matches have taken so many of the listed ones that fewer than two are left. A row of more than 65536 entries
// has no list (kBowNoList in its first slot).
__global__ __launch_bounds__(256) void bow_distance_kernel(int nrows, const BowRow* __restrict__ rows,
                                                           const uint4* __restrict__ d1,
                                                           const uint32_t* __restrict__ items2,
                                                           const uint4* __restrict__ d2,
                                                           const uint8_t* __restrict__ valid2,
                                                           uint16_t* __restrict__ dist, uint32_t* __restrict__ top) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= nrows) return;
    const BowRow row = rows[r];
    const uint4 qa = d1[2 * row.idx1], qb = d1[2 * row.idx1 + 1];
    uint32_t best[kBowTop];                         // this lane's smallest keys, ascending
#pragma unroll
    for (int t = 0; t < kBowTop; t++) best[t] = kBowListEnd;
    for (int j = lane; j < row.count2; j += 64) {
        const uint32_t i2 = items2[row.start2 + j];
        const uint4 ra = d2[2 * i2], rb = d2[2 * i2 + 1];
        const int d = __popc(qa.x ^ ra.x) + __popc(qa.y ^ ra.y) + __popc(qa.z ^ ra.z) + __popc(qa.w ^ ra.w) +
                      __popc(qb.x ^ rb.x) + __popc(qb.y ^ rb.y) + __popc(qb.z ^ rb.z) + __popc(qb.w ^ rb.w);
        dist[row.out + j] = (uint16_t)d;
        uint32_t key = (valid2 == nullptr || valid2[i2]) ? ((uint32_t)d << 16) | (uint32_t)(j & 0xFFFF) : kBowListEnd;
#pragma unroll
        for (int t = 0; t < kBowTop; t++) {
            const uint32_t lo = min(best[t], key);
            key = max(best[t], key);
            best[t] = lo;
        }
    }
#pragma unroll
    for (int t = 0; t < kBowTop; t++) {
        uint32_t m = best[0];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, o, 64));
        if (best[0] == m && m != kBowListEnd) {     // keys are distinct: exactly one lane owns the minimum
#pragma unroll
            for (int u = 0; u + 1 < kBowTop; u++) best[u] = best[u + 1];
            best[kBowTop - 1] = kBowListEnd;
        }
        if (lane == 0) top[(size_t)r * kBowTop + t] = (row.count2 > 0x10000 && t == 0) ? kBowNoList : m;
    }
}

hipError_t launch_bow_distances(int nrows, const BowRow* rows, const uint8_t* d1, const uint32_t* items2,
                                const uint8_t* d2, const uint8_t* valid2, uint16_t* dist, uint32_t* top,
                                hipStream_t s) {
    if (nrows <= 0) return hipSuccess;
    hipLaunchKernelGGL(bow_distance_kernel, dim3((nrows + 3) / 4), dim3(256), 0, s, nrows, rows,
                       reinterpret_cast<const uint4*>(d1), items2, reinterpret_cast<const uint4*>(d2), valid2, dist, top);
    return hipGetLastError();
}

// The walk over the two sorted node lists (:183-269 / :571-655: equal keys pair up, otherwise the
// lagging side jumps with lower_bound) and the rows it produces, in visiting order.
// Returns false if an index is out of range.
bool build_bow_rows(const uint32_t* nodes1, const int32_t* start1, const uint32_t* items1, int nn1,
                    const uint8_t* valid1, int n1, const uint32_t* nodes2, const int32_t* start2,
                    const uint32_t* items2, int nn2, int n2, std::vector<BowRow>& rows, long long& npairs) {
    rows.clear();
    npairs = 0;
    int a = 0, b = 0;
    while (a < nn1 && b < nn2) {
        if (nodes1[a] == nodes2[b]) {
            const int s2 = start2[b], c2 = start2[b + 1] - s2;
            if (c2 < 0 || start1[a + 1] < start1[a]) return false;
            for (int j = 0; j < c2; j++)
                if (items2[s2 + j] >= (uint32_t)n2) return false;
            for (int i = start1[a]; i < start1[a + 1]; i++) {
                const uint32_t idx1 = items1[i];
                if (idx1 >= (uint32_t)n1) return false;
                if (!valid1[idx1]) continue;                       // no map point, or a bad one (:196-200)
                if (npairs + c2 > 0x7fffffffLL) return false;
                rows.push_back(BowRow{(int32_t)idx1, s2, c2, (int32_t)npairs});
                npairs += c2;
            }
            a++;
            b++;
        } else if (nodes1[a] < nodes2[b]) {
            while (a < nn1 && nodes1[a] < nodes2[b]) a++;          // lower_bound on a sorted key list
        } else {
            while (b < nn2 && nodes2[b] < nodes1[a]) b++;
        }
    }
    return true;
}

namespace {
// ORBmatcher::ComputeThreeMaxima (src/ORBmatcher.cc:1652-1693) over the bin sizes
void bow_three_maxima(const int* histo, int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = histo[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            ind3 = ind2; ind2 = ind1; ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            ind3 = ind2; ind2 = i;
        } else if (s > max3) {
            max3 = s;
            ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) {
        ind2 = -1;
        ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
        ind3 = -1;
    }
}
}  // namespace

// Host half over the distance rows. mode 0: :190-256 and :272-290 (matches indexed by the right side,
// holding the left index); mode 1: :575-642 and :657-680 (matches indexed by the left side).
// A row's best and second-best among the right-hand features no earlier match has taken are the first two
// untaken entries of its (distance, position)-ordered list; the row itself is read only when all listed entries
// have been looked at without finding two and the list was full (so the row may hold more).
int replay_search_by_bow(int mode, const std::vector<BowRow>& rows, const uint16_t* dist, const uint32_t* top,
                         const uint32_t* items2, const orbx_keypoint* k1, const orbx_keypoint* k2,
                         const uint8_t* valid2, int n1, int n2, float nnratio, bool checkOri, int32_t* matches,
                         BowReplayScratch& tmp) {
    const int HISTO_LENGTH = 30, TH_LOW = 50;   // include/PresetParameters.h:11-13
    const int nout = mode == 0 ? n2 : n1;
    for (int i = 0; i < nout; i++) matches[i] = -1;
    if (mode == 1) tmp.matched2.assign((size_t)n2, 0);
    tmp.binOf.clear();                           // (entry of matches, bin) of every accepted match, in order
    int histo[HISTO_LENGTH] = {};
    const float factor = 1.0f / HISTO_LENGTH;
    int nmatches = 0;
    auto taken = [&](int idx2) { return mode == 0 ? matches[idx2] >= 0 : (tmp.matched2[idx2] || !valid2[idx2]); };
    for (size_t r = 0; r < rows.size(); r++) {
        const BowRow& row = rows[r];
        int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256, found = 0;
        bool whole = false;                      // the list did not settle the row
        for (int t = 0; found < 2; t++) {
            if (t == kBowTop) { whole = true; break; }               // more candidates may follow the listed ones
            const uint32_t key = top[r * kBowTop + t];
            if (key == kBowListEnd) break;                            // nothing else in the row can be matched
            if (key == kBowNoList) { whole = true; break; }
            const int idx2 = (int)items2[row.start2 + (int)(key & 0xFFFF)];
            if (taken(idx2)) continue;
            if (found == 0) {
                bestDist1 = (int)(key >> 16);
                bestIdx2 = idx2;
            } else {
                bestDist2 = (int)(key >> 16);
            }
            found++;
        }
        if (whole) {
            bestDist1 = 256; bestIdx2 = -1; bestDist2 = 256;
            for (int j = 0; j < row.count2; j++) {
                const int idx2 = (int)items2[row.start2 + j];
                if (taken(idx2)) continue;
                const int d = dist[row.out + j];
                if (d < bestDist1) {
                    bestDist2 = bestDist1;
                    bestDist1 = d;
                    bestIdx2 = idx2;
                } else if (d < bestDist2) {
                    bestDist2 = d;
                }
            }
        }
        const bool close = mode == 0 ? bestDist1 <= TH_LOW : bestDist1 < TH_LOW;   // :231 vs :619
        if (!close || !((float)bestDist1 < nnratio * (float)bestDist2)) continue;
        if (mode == 0) {
            matches[bestIdx2] = row.idx1;
        } else {
            matches[row.idx1] = bestIdx2;
            tmp.matched2[bestIdx2] = 1;
        }
        if (checkOri) {
            float rot = k1[row.idx1].angle - k2[bestIdx2].angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)std::round(rot / (360 * factor));
            if (bin == HISTO_LENGTH) bin = 0;
            if (bin >= 0 && bin < HISTO_LENGTH) {
                histo[bin]++;
                tmp.binOf.push_back({mode == 0 ? bestIdx2 : row.idx1, bin});
            }
        }
        nmatches++;
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        bow_three_maxima(histo, HISTO_LENGTH, ind1, ind2, ind3);
        for (const auto& e : tmp.binOf) {
            if (e.second == ind1 || e.second == ind2 || e.second == ind3) continue;
            matches[e.first] = -1;
            nmatches--;
        }
    }
    return nmatches;
}

}  // namespace orbx
