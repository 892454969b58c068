// ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:413-530).
//
// The data-parallel part — Frame::GetFeaturesInArea (src/Frame.cc:365-418) for every level-0 keypoint of
// frame 1 and ORBmatcher::DescriptorDistance (:1698-1714) to every candidate — runs on the GPU, one thread
// per frame-1 keypoint, and leaves per-keypoint candidate lists (index, distance) in the order the
// reference visits them. The rest of the function depends on the order in which frame-1 keypoints are
// processed (vMatchedDistance, vnMatches21), so it is replayed on the host over those lists.
#include <cmath>
#include <vector>

#include "orbx_internal.h"

namespace orbx {

struct GridParams {
    float minX, minY, invW, invH;
};

__global__ __launch_bounds__(256) void window_candidates_kernel(int n1, const orbx_keypoint* __restrict__ k1,
                                                                const uint4* __restrict__ d1,
                                                                const orbx_keypoint* __restrict__ k2,
                                                                const uint4* __restrict__ d2,
                                                                const int32_t* __restrict__ cellStart,
                                                                const int32_t* __restrict__ cellItems,
                                                                const float* __restrict__ prev, GridParams gp, float r,
                                                                int maxCand, int32_t* __restrict__ candCount,
                                                                int32_t* __restrict__ candIdx,
                                                                int32_t* __restrict__ candDist) {
    const int i1 = blockIdx.x * 256 + threadIdx.x;
    if (i1 >= n1) return;
    int count = 0;
    if (k1[i1].octave <= 0) {                                  // level1 > 0: continue (:429-431)
        const float x = prev[2 * i1], y = prev[2 * i1 + 1];
        // GetFeaturesInArea(x, y, r, 0, 0)
        const int nMinCellX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, gp.minX), r), gp.invW)));
        const int nMaxCellX = min(ORBX_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, gp.minX), r), gp.invW)));
        const int nMinCellY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, gp.minY), r), gp.invH)));
        const int nMaxCellY = min(ORBX_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, gp.minY), r), gp.invH)));
        if (!(nMinCellX >= ORBX_GRID_COLS || nMaxCellX < 0 || nMinCellY >= ORBX_GRID_ROWS || nMaxCellY < 0)) {
            const uint4 qa = d1[2 * i1], qb = d1[2 * i1 + 1];
            for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
                for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                    const int c = ix * ORBX_GRID_ROWS + iy;
                    for (int j = cellStart[c]; j < cellStart[c + 1]; j++) {
                        const int i2 = cellItems[j];
                        const orbx_keypoint kp = k2[i2];
                        if (kp.octave < 0 || kp.octave > 0) continue;          // minLevel = maxLevel = 0
                        const float distx = __fsub_rn(kp.x, x), disty = __fsub_rn(kp.y, y);
                        if (!(fabsf(distx) < r && fabsf(disty) < r)) continue;
                        const uint4 ra = d2[2 * i2], rb = d2[2 * i2 + 1];
                        const int dist = __popc(qa.x ^ ra.x) + __popc(qa.y ^ ra.y) + __popc(qa.z ^ ra.z) +
                                         __popc(qa.w ^ ra.w) + __popc(qb.x ^ rb.x) + __popc(qb.y ^ rb.y) +
                                         __popc(qb.z ^ rb.z) + __popc(qb.w ^ rb.w);
                        if (count < maxCand) {
                            candIdx[(long long)i1 * maxCand + count] = i2;
                            candDist[(long long)i1 * maxCand + count] = dist;
                        }
                        count++;
                    }
                }
        }
    }
    candCount[i1] = count;
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

// Host half: src/ORBmatcher.cc:416-529 over the GPU-made candidate lists.
int replay_search_for_initialization(const orbx_keypoint* k1, int n1, const orbx_keypoint* k2, int n2,
                                     const int32_t* candCount, const int32_t* candIdx, const int32_t* candDist,
                                     int maxCand, float* prev, float nnratio, bool checkOri, int32_t* matches12) {
    const int HISTO_LENGTH = 30, TH_LOW = 50;   // include/PresetParameters.h:11-13
    int nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> vMatchedDistance(n2, INT32_MAX), vnMatches21(n2, -1);
    for (int i1 = 0; i1 < n1; i1++) {
        const int nc = candCount[i1];
        if (nc == 0) continue;
        int bestDist = INT32_MAX, bestDist2 = INT32_MAX, bestIdx2 = -1;
        for (int c = 0; c < nc; c++) {
            const int i2 = candIdx[(size_t)i1 * maxCand + c], dist = candDist[(size_t)i1 * maxCand + c];
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestIdx2 = i2;
            } else if (dist < bestDist2) {
                bestDist2 = dist;
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
                    if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(i1);
                }
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx1 : rotHist[i])
                if (matches12[idx1] >= 0) {
                    matches12[idx1] = -1;
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

hipError_t launch_window_candidates(int n1, const orbx_keypoint* k1, const uint8_t* d1, const orbx_keypoint* k2,
                                    const uint8_t* d2, const int32_t* cellStart, const int32_t* cellItems,
                                    const float* prev, float minX, float minY, float invW, float invH, float r,
                                    int maxCand, int32_t* candCount, int32_t* candIdx, int32_t* candDist,
                                    hipStream_t s) {
    if (n1 <= 0) return hipSuccess;
    GridParams gp{minX, minY, invW, invH};
    hipLaunchKernelGGL(window_candidates_kernel, dim3((n1 + 255) / 256), dim3(256), 0, s, n1, k1,
                       reinterpret_cast<const uint4*>(d1), k2, reinterpret_cast<const uint4*>(d2), cellStart,
                       cellItems, prev, gp, r, maxCand, candCount, candIdx, candDist);
    return hipGetLastError();
}

}  // namespace orbx
