// ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:413-530) and
// ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) (:1372-1516) and
// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) (:41-123): the windowed searches over the
// current frame's 64 x 48 grid.
//
// The data-parallel part — Frame::GetFeaturesInArea (src/Frame.cc:365-418) for every query and
// ORBmatcher::DescriptorDistance (:1698-1714) to every candidate — runs on the GPU and leaves per-query candidate
// lists (index, distance) in the order the reference visits them. The rest of each function depends on the
// order in which the queries are processed (vMatchedDistance, vnMatches21, what a feature already holds), so it is
// replayed on the host. Two kernels: window_candidates_kernel (one wave per query, packed lists returned whole;
// the two projection searches) and window_top_kernel (one workgroup per query, lists kept on the device, an
// ordered short list per query returned; the initialization search).
#include <algorithm>
#include <cmath>
#include <functional>
#include <vector>

#include "orbx_internal.h"

namespace orbx {

struct GridParams {
    float minX, minY, invW, invH;
};

// A candidate is one word: distance << 23 | index in frame 2. A query's candidates are contiguous at
// cand[candOffset[i1] ..) in visiting order; the regions are handed out by an atomic cursor after a counting pass,
// so the list is as long as the candidates that exist and not n1 x (largest possible list). A query whose list
// does not fit below capTotal reports the count -1. queries, candCount, candOffset and cand may be pinned host
// memory (the kernel then reads its few inputs and leaves its results where the host replay picks them up, with
// no copy in between); the cursor must be device memory.
//
// One wave per frame-1 keypoint. The window's cells, numbered in the reference's visiting order (columns outside,
// rows inside, :387-410), are taken 64 at a time, one cell per lane, so the dependent loads of a cell (its start,
// its keypoint indices, their positions) run side by side for 64 cells; a prefix sum over the lanes' counts puts
// every cell's candidates where the sequential walk would have put them.
__global__ __launch_bounds__(256) void window_candidates_kernel(int n1, const WindowQuery* __restrict__ queries,
                                                                const uint4* __restrict__ d1,
                                                                const orbx_keypoint* __restrict__ k2,
                                                                const uint4* __restrict__ d2,
                                                                const float* __restrict__ uRight2,
                                                                const int32_t* __restrict__ cellStart,
                                                                const int32_t* __restrict__ cellItems, GridParams gp,
                                                                int n2, int capTotal, int32_t* __restrict__ cursor,
                                                                int32_t* __restrict__ candCount,
                                                                int32_t* __restrict__ candOffset,
                                                                uint32_t* __restrict__ cand) {
    const int lane = threadIdx.x & 63;
    const int i1 = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (i1 >= n1) return;
    int total = 0, offset = 0;
    const WindowQuery q = queries[i1];
    if (q.r >= 0.0f) {                                         // r < 0: the caller skips this query
        const float x = q.x, y = q.y, r = q.r;
        // GetFeaturesInArea(x, y, r, minLevel, maxLevel)
        const int nMinCellX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, gp.minX), r), gp.invW)));
        const int nMaxCellX = min(ORBX_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, gp.minX), r), gp.invW)));
        const int nMinCellY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, gp.minY), r), gp.invH)));
        const int nMaxCellY = min(ORBX_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, gp.minY), r), gp.invH)));
        if (!(nMinCellX >= ORBX_GRID_COLS || nMaxCellX < 0 || nMinCellY >= ORBX_GRID_ROWS || nMaxCellY < 0) &&
            nMaxCellX >= nMinCellX && nMaxCellY >= nMinCellY) {
            const int rows = nMaxCellY - nMinCellY + 1, ncells = (nMaxCellX - nMinCellX + 1) * rows;
            // this lane's cell of a chunk: [begin, end) in cellItems, empty past the window
            auto cell_range = [&](int chunk, int& jb, int& je) {
                const int w = chunk + lane;
                jb = je = 0;
                if (w < ncells) {
                    const int cx = w / rows, c = (nMinCellX + cx) * ORBX_GRID_ROWS + nMinCellY + (w - cx * rows);
                    // a grid lists a keypoint at most once: whatever the arrays hold, the walk stays inside them
                    jb = max(cellStart[c], 0);
                    je = min(cellStart[c + 1], n2);
                }
            };
            auto passes = [&](int i2) {
                if ((unsigned)i2 >= (unsigned)n2) return false;
                const orbx_keypoint kp = k2[i2];
                // (src/Frame.cc:399-406; with bCheckLevels false both limits are below every octave)
                if (kp.octave < q.minLevel || (q.maxLevel >= 0 && kp.octave > q.maxLevel)) return false;
                const float distx = __fsub_rn(kp.x, x), disty = __fsub_rn(kp.y, y);
                if (!(fabsf(distx) < r && fabsf(disty) < r)) return false;
                // stereo consistency of SearchByProjection (src/ORBmatcher.cc:1451-1457); it does not depend on
                // the order of the queries, unlike the test on what the feature already holds
                if (uRight2 != nullptr) {
                    const float ur2 = uRight2[i2];
                    if (ur2 > 0 && fabsf(__fsub_rn(q.ur, ur2)) > r) return false;
                }
                return true;
            };
            auto wave_incl_scan = [&](int v) {
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int u = __shfl_up(v, off);
                    if (lane >= off) v += u;
                }
                return v;
            };
            for (int chunk = 0; chunk < ncells; chunk += 64) {
                int jb, je, cnt = 0;
                cell_range(chunk, jb, je);
                for (int j = jb; j < je; j++) cnt += passes(cellItems[j]);
                total += __builtin_amdgcn_readlane(wave_incl_scan(cnt), 63);
            }
            if (total > 0) {
                if (lane == 0) offset = atomicAdd(cursor, total);
                offset = __builtin_amdgcn_readfirstlane(offset);
                if (offset + total <= capTotal) {
                    const uint4 qa = d1[2 * i1], qb = d1[2 * i1 + 1];
                    int done = offset;
                    for (int chunk = 0; chunk < ncells; chunk += 64) {
                        int jb, je, cnt = 0;
                        cell_range(chunk, jb, je);
                        for (int j = jb; j < je; j++) cnt += passes(cellItems[j]);
                        const int incl = wave_incl_scan(cnt);
                        int k = done + incl - cnt;
                        for (int j = jb; j < je; j++) {
                            const int i2 = cellItems[j];
                            if (!passes(i2)) continue;
                            const uint4 ra = d2[2 * i2], rb = d2[2 * i2 + 1];
                            const int dist = __popc(qa.x ^ ra.x) + __popc(qa.y ^ ra.y) + __popc(qa.z ^ ra.z) +
                                             __popc(qa.w ^ ra.w) + __popc(qb.x ^ rb.x) + __popc(qb.y ^ rb.y) +
                                             __popc(qb.z ^ rb.z) + __popc(qb.w ^ rb.w);
                            cand[k++] = ((uint32_t)dist << 23) | (uint32_t)i2;
                        }
                        done += __builtin_amdgcn_readlane(incl, 63);
                    }
                } else {
                    total = -1;                            // no room left in the list: the caller runs again with more
                }
            }
        }
    }
    if (lane == 0) {
        candCount[i1] = total;
        candOffset[i1] = offset;
    }
}

// The same walk for a caller whose host pass needs little of the lists (SearchForInitialization). One workgroup
// of kWinThreads lanes per active query, so the cells of an ordinary window are all visited at once and the
// dependent loads of the walk (cell start -> keypoint index -> keypoint -> descriptor) are paid once rather than
// once per 64 cells. Query a keeps its whole list in device memory at cand[a * n2 ..) in visiting order (a grid
// lists a keypoint at most once, so n2 entries always suffice; arrays that are not a grid lose what does not fit),
// and what travels to the host is a record of kWinStride words: {count, the kWinTop entries smallest by
// (distance, visiting position) as distance << 23 | index, closed by kWinListEnd}.
__global__ __launch_bounds__(kWinThreads) void window_top_kernel(const WindowQuery* __restrict__ queries,
                                                                 const int32_t* __restrict__ queryIndex,
                                                                 const uint4* __restrict__ d1,
                                                                 const orbx_keypoint* __restrict__ k2,
                                                                 const uint4* __restrict__ d2,
                                                                 const int32_t* __restrict__ cellStart,
                                                                 const int32_t* __restrict__ cellItems, GridParams gp,
                                                                 int n2, uint32_t* __restrict__ cand,
                                                                 uint32_t* __restrict__ out) {
    constexpr int kWaves = kWinThreads / 64;
    static_assert(kWaves * kWinTop == 64, "the final merge gives every lane of one wave one entry");
    __shared__ int waveSum[kWaves];
    __shared__ uint32_t topKey[64], topPay[64];
    const int a = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const WindowQuery q = queries[a];
    const int i1 = queryIndex[a];
    uint32_t* list = cand + (size_t)a * (size_t)n2;
    int count = 0;
    uint32_t best[kWinTop], pay[kWinTop];                      // this lane's smallest keys, ascending, and their indices
#pragma unroll
    for (int t = 0; t < kWinTop; t++) { best[t] = kWinListEnd; pay[t] = 0; }
    const float x = q.x, y = q.y, r = q.r;
    // GetFeaturesInArea(x, y, r, minLevel, maxLevel)
    const int nMinCellX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, gp.minX), r), gp.invW)));
    const int nMaxCellX = min(ORBX_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, gp.minX), r), gp.invW)));
    const int nMinCellY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, gp.minY), r), gp.invH)));
    const int nMaxCellY = min(ORBX_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, gp.minY), r), gp.invH)));
    const bool inside = r >= 0.0f && !(nMinCellX >= ORBX_GRID_COLS || nMaxCellX < 0 || nMinCellY >= ORBX_GRID_ROWS || nMaxCellY < 0) &&
                        nMaxCellX >= nMinCellX && nMaxCellY >= nMinCellY;
    const int rows = nMaxCellY - nMinCellY + 1, ncells = inside ? (nMaxCellX - nMinCellX + 1) * rows : 0;
    auto passes = [&](int i2) {
        if ((unsigned)i2 >= (unsigned)n2) return false;
        const orbx_keypoint kp = k2[i2];
        if (kp.octave < q.minLevel || (q.maxLevel >= 0 && kp.octave > q.maxLevel)) return false;
        const float distx = __fsub_rn(kp.x, x), disty = __fsub_rn(kp.y, y);
        return fabsf(distx) < r && fabsf(disty) < r;
    };
    uint4 qa = {0, 0, 0, 0}, qb = {0, 0, 0, 0};
    if (ncells > 0) {
        qa = d1[2 * i1];
        qb = d1[2 * i1 + 1];
    }
    for (int chunk = 0; chunk < ncells; chunk += kWinThreads) {            // trip count uniform over the workgroup
        const int w = chunk + tid;
        int jb = 0, je = 0, cnt = 0;
        if (w < ncells) {
            const int cx = w / rows, c = (nMinCellX + cx) * ORBX_GRID_ROWS + nMinCellY + (w - cx * rows);
            jb = max(cellStart[c], 0);
            je = min(cellStart[c + 1], n2);
        }
        for (int j = jb; j < je; j++) cnt += passes(cellItems[j]);
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int u = __shfl_up(incl, off);
            if (lane >= off) incl += u;
        }
        if (lane == 63) waveSum[wave] = incl;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int v = 0; v < kWaves; v++) {
            const int sum = waveSum[v];
            before += v < wave ? sum : 0;
            total += sum;
        }
        int k = count + before + incl - cnt;
        for (int j = jb; j < je; j++) {
            const int i2 = cellItems[j];
            if (!passes(i2)) continue;
            if (k >= n2) break;
            const uint4 ra = d2[2 * i2], rb = d2[2 * i2 + 1];
            const int dist = __popc(qa.x ^ ra.x) + __popc(qa.y ^ ra.y) + __popc(qa.z ^ ra.z) + __popc(qa.w ^ ra.w) +
                             __popc(qb.x ^ rb.x) + __popc(qb.y ^ rb.y) + __popc(qb.z ^ rb.z) + __popc(qb.w ^ rb.w);
            list[k] = ((uint32_t)dist << 23) | (uint32_t)i2;
            uint32_t key = ((uint32_t)dist << 23) | (uint32_t)k, p = (uint32_t)i2;
#pragma unroll
            for (int t = 0; t < kWinTop; t++) {
                const bool less = key < best[t];
                const uint32_t kk = less ? best[t] : key, pp = less ? pay[t] : p;
                best[t] = less ? key : best[t];
                pay[t] = less ? p : pay[t];
                key = kk;
                p = pp;
            }
            k++;
        }
        count = min(count + total, n2);
        __syncthreads();                                       // waveSum is written again in the next round
    }
    auto wave_min = [](uint32_t m) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, o, 64));
        return m;
    };
    // every wave's kWinTop smallest, then the smallest of those; positions differ, so a minimum has one owner
#pragma unroll
    for (int t = 0; t < kWinTop; t++) {
        const uint32_t m = wave_min(best[0]);
        if (lane == 0) topKey[wave * kWinTop + t] = m;
        if (best[0] == m && m != kWinListEnd) {
            topPay[wave * kWinTop + t] = pay[0];
#pragma unroll
            for (int u = 0; u + 1 < kWinTop; u++) { best[u] = best[u + 1]; pay[u] = pay[u + 1]; }
            best[kWinTop - 1] = kWinListEnd;
        }
    }
    __syncthreads();
    if (wave == 0) {
        uint32_t* rec = out + (size_t)a * kWinStride;
        uint32_t key = topKey[lane];
        const uint32_t p = topPay[lane];
#pragma unroll
        for (int t = 0; t < kWinTop; t++) {
            const uint32_t m = wave_min(key);
            if (m == kWinListEnd) {
                if (lane == 0) rec[1 + t] = kWinListEnd;
            } else if (key == m) {
                rec[1 + t] = (m & 0xFF800000u) | p;
                key = kWinListEnd;
            }
        }
        if (lane == 0) rec[0] = (uint32_t)count;
    }
}

hipError_t launch_window_top(int nq, const WindowQuery* queries, const int32_t* queryIndex, const uint8_t* d1,
                             const orbx_keypoint* k2, const uint8_t* d2, const int32_t* cellStart,
                             const int32_t* cellItems, float minX, float minY, float invW, float invH, int n2,
                             uint32_t* cand, uint32_t* out, hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    GridParams gp{minX, minY, invW, invH};
    hipLaunchKernelGGL(window_top_kernel, dim3(nq), dim3(kWinThreads), 0, s, queries, queryIndex,
                       reinterpret_cast<const uint4*>(d1), k2, reinterpret_cast<const uint4*>(d2), cellStart, cellItems,
                       gp, n2, cand, out);
    return hipGetLastError();
}

namespace {
// ORBmatcher::ComputeThreeMaxima (src/ORBmatcher.cc:1652-1693)
void three_maxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
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

// Host half of SearchForInitialization (src/ORBmatcher.cc:416-529) over the per-query records of
// window_top_kernel. The best and second-best of a query among the
// candidates the reference would not skip (:455, vMatchedDistance[i2] <= dist) are the first two unskipped entries
// of its ordered list. When the list ends before the query's candidates do, everything not listed is at least as
// far as the last listed entry, which settles most queries anyway: no match is possible once that distance is
// above TH_LOW and nothing unskipped was listed, and with one unskipped entry the ratio test against the last
// listed distance is the weakest it can be. Only what is left reads the query's whole list (fetch(query, count)).
int replay_search_for_initialization_top(const orbx_keypoint* k1, int n1, const orbx_keypoint* k2, int n2, int nq,
                                         const int32_t* queryIndex, const uint32_t* records,
                                         const std::function<const uint32_t*(int, int)>& fetch, float* prev,
                                         float nnratio, bool checkOri, int32_t* matches12, InitReplayScratch& tmp) {
    const int HISTO_LENGTH = 30, TH_LOW = 50;   // include/PresetParameters.h:11-13
    int nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    int histo[HISTO_LENGTH] = {};
    tmp.binOf.clear();
    const float factor = 1.0f / HISTO_LENGTH;
    tmp.matchedDistance.assign((size_t)n2, INT32_MAX);
    tmp.matches21.assign((size_t)n2, -1);
    std::vector<int>& vMatchedDistance = tmp.matchedDistance;
    std::vector<int>& vnMatches21 = tmp.matches21;
    for (int a = 0; a < nq; a++) {                 // the frame-1 keypoints the reference does not skip, in its order
        const int i1 = queryIndex[a];
        const uint32_t* rec = records + (size_t)a * kWinStride;
        const int nc = (int)rec[0];
        if (nc == 0) continue;
        if (nc < 0 || nc > n2) return -1;
        int bestDist = INT32_MAX, bestDist2 = INT32_MAX, bestIdx2 = -1, found = 0;
        const int listed = std::min(nc, kWinTop);
        bool whole = false;
        for (int t = 0; t < listed && found < 2; t++) {
            const uint32_t w = rec[1 + t];
            if (w == kWinListEnd) { whole = true; break; }          // (a record never ends early; be safe)
            const int i2 = (int)(w & 0x7FFFFFu), dist = (int)(w >> 23);
            if (i2 >= n2) return -1;
            if (vMatchedDistance[i2] <= dist) continue;
            if (found == 0) {
                bestDist = dist;
                bestIdx2 = i2;
                if (dist > TH_LOW) break;                           // nothing closer exists: no match
            } else {
                bestDist2 = dist;
            }
            found++;
        }
        if (!whole && found < 2 && nc > listed && !(found == 0 && bestDist != INT32_MAX)) {
            const int lastDist = (int)(rec[1 + listed - 1] >> 23);  // every unlisted candidate is at least this far
            if (found == 0)
                whole = lastDist <= TH_LOW;
            else
                whole = !(nnratio >= 0.0f && (float)bestDist < (float)lastDist * nnratio);
        }
        if (whole) {
            const uint32_t* list = fetch(a, nc);
            if (!list) return -1;
            bestDist = INT32_MAX; bestDist2 = INT32_MAX; bestIdx2 = -1;
            for (int c = 0; c < nc; c++) {
                const int i2 = (int)(list[c] & 0x7FFFFFu), dist = (int)(list[c] >> 23);
                if (i2 >= n2) return -1;
                if (vMatchedDistance[i2] <= dist) continue;
                if (dist < bestDist) {
                    bestDist2 = bestDist;
                    bestDist = dist;
                    bestIdx2 = i2;
                } else if (dist < bestDist2) {
                    bestDist2 = dist;
                }
            }
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (vnMatches21[bestIdx2] >= 0) {
                    matches12[vnMatches21[bestIdx2]] = -1;
                    nmatches--;
                }
                matches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (checkOri) {
                    float rot = k1[i1].angle - k2[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot / (360 * factor));
                    if (bin == HISTO_LENGTH) bin = 0;
                    if (bin >= 0 && bin < HISTO_LENGTH) {
                        histo[bin]++;
                        tmp.binOf.push_back({i1, bin});
                    }
                }
            }
        }
    }
    if (checkOri) {
        // ComputeThreeMaxima (:1652-1693) over the bin sizes
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) {
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
        for (const auto& e : tmp.binOf) {
            if (e.second == ind1 || e.second == ind2 || e.second == ind3) continue;
            if (matches12[e.first] >= 0) {
                matches12[e.first] = -1;
                nmatches--;
            }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)
        if (matches12[i1] >= 0) {
            prev[2 * i1] = k2[matches12[i1]].x;
            prev[2 * i1 + 1] = k2[matches12[i1]].y;
        }
    return nmatches;
}

hipError_t launch_window_candidates(int n1, const WindowQuery* queries, const uint8_t* d1, const orbx_keypoint* k2,
                                    const uint8_t* d2, const float* uRight2, const int32_t* cellStart,
                                    const int32_t* cellItems, float minX, float minY, float invW, float invH,
                                    int n2, int capTotal, int32_t* cursor, int32_t* candCount, int32_t* candOffset,
                                    uint32_t* cand, hipStream_t s) {
    if (n1 <= 0) return hipSuccess;
    GridParams gp{minX, minY, invW, invH};
    hipError_t e = hipMemsetAsync(cursor, 0, sizeof(int32_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(window_candidates_kernel, dim3((n1 + 3) / 4), dim3(256), 0, s, n1, queries,
                       reinterpret_cast<const uint4*>(d1), k2, reinterpret_cast<const uint4*>(d2), uRight2, cellStart,
                       cellItems, gp, n2, capTotal, cursor, candCount, candOffset, cand);
    return hipGetLastError();
}

// Host half of SearchByProjection(Frame&, const Frame&, ...): src/ORBmatcher.cc:1441-1513 over the GPU-made
// candidate lists (level, window and stereo tests already applied). What a feature of the current frame holds
// changes as the loop goes, so the best match of a map point depends on the points before it.
int replay_search_by_projection_frame(int npts, const int32_t* candCount, const int32_t* candOffset,
                                      const uint32_t* cand, const float* lastAngle, const uint8_t* hasObs,
                                      const orbx_keypoint* k2, const uint8_t* occupied2, int n2, int maxDist,
                                      bool checkOri, int32_t* match2) {
    const int HISTO_LENGTH = 30;                  // include/PresetParameters.h:11-13
    std::vector<uint8_t> blocked(n2, 0);          // mvpMapPoints[i2] && mvpMapPoints[i2]->Observations() > 0
    for (int i = 0; i < n2; i++) {
        match2[i] = -1;
        blocked[i] = occupied2 ? (occupied2[i] != 0) : 0;
    }
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = 0; i < npts; i++) {
        const int nc = candCount[i];
        if (nc == 0) continue;
        const uint32_t* list = cand + candOffset[i];
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < nc; c++) {
            const int i2 = (int)(list[c] & 0x7FFFFFu), dist = (int)(list[c] >> 23);
            if (blocked[i2]) continue;
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx2 = i2;
            }
        }
        if (bestDist <= maxDist) {                // TH_HIGH (:1470) or ORBdist (:1604)
            match2[bestIdx2] = i;
            blocked[bestIdx2] = hasObs[i] != 0;
            nmatches++;
            if (checkOri) {
                float rot = lastAngle[i] - k2[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot / (360 * factor));
                if (bin == HISTO_LENGTH) bin = 0;
                if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx2 : rotHist[i]) {         // unconditional, as in the reference (:1506-1510)
                match2[idx2] = -1;
                nmatches--;
            }
        }
    }
    return nmatches;
}

// Host half of SearchByProjection(Frame&, const vector<MapPoint*>&, th): src/ORBmatcher.cc:70-119 over the
// GPU-made candidate lists.
int replay_search_by_projection_points(int npts, const int32_t* candCount, const int32_t* candOffset,
                                       const uint32_t* cand, const uint8_t* hasObs, const orbx_keypoint* k2,
                                       const uint8_t* occupied2, int n2, float nnratio, int32_t* match2) {
    const int TH_HIGH = 100;
    std::vector<uint8_t> blocked(n2, 0);
    for (int i = 0; i < n2; i++) {
        match2[i] = -1;
        blocked[i] = occupied2 ? (occupied2[i] != 0) : 0;
    }
    int nmatches = 0;
    for (int i = 0; i < npts; i++) {
        const int nc = candCount[i];
        if (nc == 0) continue;
        const uint32_t* list = cand + candOffset[i];
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = (int)(list[c] & 0x7FFFFFu), dist = (int)(list[c] >> 23);
            if (blocked[idx]) continue;
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestLevel2 = bestLevel;
                bestLevel = k2[idx].octave;
                bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = k2[idx].octave;
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {
            // ratio to the second match only when both sit on the same pyramid level (:114)
            if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
            match2[bestIdx] = i;
            blocked[bestIdx] = hasObs[i] != 0;
            nmatches++;
        }
    }
    return nmatches;
}

}  // namespace orbx
