// 256-bit Hamming matching: ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1698-1714), the
// best / second-best scan that every Search* routine wraps around it (canonical form at
// src/ORBmatcher.cc:204-229) and the descriptor half of Frame::ComputeStereoMatches
// (src/Frame.cc:503-586).
//
// kNN kernel: one thread owns one query descriptor in 8 VGPRs. All 64 lanes of a wave walk the same
// database row at the same time, so the row address is wave-uniform and the row arrives through
// scalar loads into SGPRs; one pair costs 8 v_xor_b32 + 8 v_bcnt_u32_b32 (the popcount
// instruction adds its accumulator for free) and three more VALU ops for the running best two.
// The two best are kept as 32-bit keys (distance << 23 | row offset inside the chunk): taking
// min / median-of-3 on keys reproduces the reference's strict `<` updates, i.e. the lowest row
// index wins ties for the best and the second best is the second order statistic of the distances.
// No MFMA: this is integer bit-counting, not a dense float contraction.
#include "orbx_internal.h"

namespace orbx {

namespace {

constexpr int kQueriesPerBlock = 256;
constexpr int64_t kMaxChunkRows = 1 << 22;   // row offset must fit the 23-bit key field
constexpr unsigned kEmptyKey = 0xFFFFFFFFu;

struct Partial {       // result of one query over one database chunk
    int32_t d1, d2;    // distances, or INT32_MAX when the chunk had fewer than 1 / 2 rows
    int64_t idx;       // global row of d1, -1 if none
};

__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c) {
    return max(min(a, b), min(max(a, b), c));
}

}  // namespace

__global__ __launch_bounds__(kQueriesPerBlock) void knn2_kernel(const uint4* __restrict__ query, int64_t nq,
                                                                const uint4* __restrict__ db, int64_t ndb,
                                                                int64_t chunkRows, int64_t indexBase,
                                                                Partial* __restrict__ partial) {
    const int64_t qi = (int64_t)blockIdx.x * kQueriesPerBlock + threadIdx.x;
    const int64_t qc = qi < nq ? qi : nq - 1;
    const uint4 qa = query[2 * qc], qb = query[2 * qc + 1];
    const int64_t j0 = (int64_t)blockIdx.y * chunkRows;
    const int rows = (int)min(chunkRows, ndb - j0);
    const uint4* __restrict__ base = db + 2 * j0;
    unsigned k1 = kEmptyKey, k2 = kEmptyKey;
#pragma unroll 8
    for (int j = 0; j < rows; j++) {
        const uint4 ra = base[2 * j], rb = base[2 * j + 1];   // wave-uniform address: scalar loads
        int d = __popc(qa.x ^ ra.x);
        d += __popc(qa.y ^ ra.y);
        d += __popc(qa.z ^ ra.z);
        d += __popc(qa.w ^ ra.w);
        d += __popc(qb.x ^ rb.x);
        d += __popc(qb.y ^ rb.y);
        d += __popc(qb.z ^ rb.z);
        d += __popc(qb.w ^ rb.w);
        const unsigned key = ((unsigned)d << 23) | (unsigned)j;
        k2 = umed3(k1, k2, key);   // k1 <= k2 always holds
        k1 = min(k1, key);
    }
    if (qi >= nq) return;
    Partial p;
    p.d1 = k1 == kEmptyKey ? INT32_MAX : (int32_t)(k1 >> 23);
    p.d2 = k2 == kEmptyKey ? INT32_MAX : (int32_t)(k2 >> 23);
    p.idx = k1 == kEmptyKey ? -1 : indexBase + j0 + (int64_t)(k1 & 0x7FFFFFu);
    partial[(int64_t)blockIdx.y * nq + qi] = p;
}

// Folds the per-chunk partials of one query in chunk (= index) order with the reference's update
// rule, then applies the start value `init`: a distance that is not < init never replaces it.
__global__ __launch_bounds__(256) void knn2_fold_kernel(const Partial* __restrict__ partial, int nchunks,
                                                        int64_t nq, int32_t init, int64_t* __restrict__ idx,
                                                        int32_t* __restrict__ d1, int32_t* __restrict__ d2) {
    const int64_t qi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    int32_t b1 = init, b2 = init;
    int64_t bi = -1;
    for (int c = 0; c < nchunks; c++) {
        const Partial p = partial[(int64_t)c * nq + qi];
        if (p.d1 < b1) { b2 = b1; b1 = p.d1; bi = p.idx; }
        else if (p.d1 < b2) b2 = p.d1;
        if (p.d2 < b2) b2 = p.d2;
    }
    idx[qi] = bi;
    d1[qi] = b1;
    d2[qi] = b2;
}

// Same fold over results that came from separate scans of consecutive database shards.
__global__ __launch_bounds__(256) void knn2_merge_kernel(int nparts, int64_t nq, const int64_t* __restrict__ idx,
                                                         const int32_t* __restrict__ d1,
                                                         const int32_t* __restrict__ d2, int32_t init,
                                                         int64_t* __restrict__ oidx, int32_t* __restrict__ od1,
                                                         int32_t* __restrict__ od2) {
    const int64_t qi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    int32_t b1 = init, b2 = init;
    int64_t bi = -1;
    for (int p = 0; p < nparts; p++) {
        const int32_t a1 = d1[(int64_t)p * nq + qi], a2 = d2[(int64_t)p * nq + qi];
        const int64_t ai = idx[(int64_t)p * nq + qi];
        if (ai >= 0 && a1 < b1) { b2 = b1; b1 = a1; bi = ai; }
        else if (ai >= 0 && a1 < b2) b2 = a1;
        if (a2 < b2) b2 = a2;
    }
    oidx[qi] = bi;
    od1[qi] = b1;
    od2[qi] = b2;
}

// bestDist <= TH (src/ORBmatcher.cc:231,467) or < TH (:619), then the float ratio test (:233,469,621)
__global__ __launch_bounds__(256) void ratio_test_kernel(int64_t nq, const int32_t* __restrict__ d1,
                                                         const int32_t* __restrict__ d2, int32_t th, int inclusive,
                                                         float ratio, uint8_t* __restrict__ accept) {
    const int64_t qi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    const int a = d1[qi], b = d2[qi];
    const bool ok = (inclusive ? a <= th : a < th) && (float)a < __fmul_rn(ratio, (float)b);
    accept[qi] = ok ? 1 : 0;
}

// One thread per left keypoint walks every right keypoint of its pair in index order; the row
// table of the reference (:513-530) becomes the test minr <= (int)vL <= maxr on each right point.
__global__ __launch_bounds__(256) void stereo_band_kernel(int cap, const orbx_keypoint* __restrict__ kl,
                                                          const uint4* __restrict__ dl, const int32_t* __restrict__ nl,
                                                          const orbx_keypoint* __restrict__ kr,
                                                          const uint4* __restrict__ dr, const int32_t* __restrict__ nr,
                                                          const float* __restrict__ scale, int nlevels, float maxD,
                                                          int32_t* __restrict__ bestIdx, int32_t* __restrict__ bestDist) {
    const int pair = blockIdx.y;
    const int iL = blockIdx.x * 256 + threadIdx.x;
    const int nL = min(nl[pair], cap), nR = min(nr[pair], cap);
    if (iL >= nL) return;
    const long long o = (long long)pair * cap;
    const orbx_keypoint L = kl[o + iL];
    const uint4 qa = dl[2 * (o + iL)], qb = dl[2 * (o + iL) + 1];
    const int row = (int)L.y;                       // vRowIndices[vL], float -> index truncation (:548)
    const float minU = __fsub_rn(L.x, maxD), maxU = L.x;   // minD = 0 (:534,553-554)
    int best = 100, bi = -1;                        // TH_HIGH (:559)
    if (maxU >= 0) {                                // (:556)
        for (int iR = 0; iR < nR; iR++) {
            const orbx_keypoint R = kr[o + iR];     // wave-uniform address
            const int oct = min(max(R.octave, 0), nlevels - 1);
            const float r = __fmul_rn(2.0f, scale[oct]);
            const int maxr = (int)ceilf(__fadd_rn(R.y, r)), minr = (int)floorf(__fsub_rn(R.y, r));
            if (row < minr || row > maxr) continue;
            if (R.octave < L.octave - 1 || R.octave > L.octave + 1) continue;   // (:570)
            if (!(R.x >= minU && R.x <= maxU)) continue;                        // (:575)
            const uint4 ra = dr[2 * (o + iR)], rb = dr[2 * (o + iR) + 1];
            const int d = __popc(qa.x ^ ra.x) + __popc(qa.y ^ ra.y) + __popc(qa.z ^ ra.z) + __popc(qa.w ^ ra.w) +
                          __popc(qb.x ^ rb.x) + __popc(qb.y ^ rb.y) + __popc(qb.z ^ rb.z) + __popc(qb.w ^ rb.w);
            if (d < best) { best = d; bi = iR; }    // strict: lowest right index wins ties (:580)
        }
    }
    bestIdx[o + iL] = bi;
    bestDist[o + iL] = best;
}

namespace {
inline int64_t pick_chunk_rows(int64_t nq, int64_t ndb) {
    // enough (query block x chunk) workgroups to fill 256 CUs several times over
    const int64_t qblocks = (nq + kQueriesPerBlock - 1) / kQueriesPerBlock;
    int64_t want = (4096 + qblocks - 1) / qblocks;          // chunks wanted
    if (want < 1) want = 1;
    int64_t rows = (ndb + want - 1) / want;
    if (rows < 2048) rows = 2048;                            // keep the fold cheap
    if (rows > kMaxChunkRows) rows = kMaxChunkRows;
    return rows;
}
}  // namespace

size_t knn2_scratch_bytes(int64_t nq, int64_t ndb) {
    if (nq <= 0) return 0;
    const int64_t rows = pick_chunk_rows(nq, ndb);
    const int64_t nchunks = ndb > 0 ? (ndb + rows - 1) / rows : 1;
    return (size_t)nchunks * (size_t)nq * sizeof(Partial);
}

hipError_t launch_knn2(const uint8_t* q, int64_t nq, const uint8_t* db, int64_t ndb, int32_t init,
                       int64_t indexBase, int64_t* idx, int32_t* d1, int32_t* d2, void* scratch,
                       size_t scratchBytes, hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    const int64_t rows = pick_chunk_rows(nq, ndb);
    const int64_t nchunks = ndb > 0 ? (ndb + rows - 1) / rows : 0;
    if ((size_t)nchunks * (size_t)nq * sizeof(Partial) > scratchBytes) return hipErrorInvalidValue;
    if (nchunks > 65535) return hipErrorInvalidValue;
    Partial* part = static_cast<Partial*>(scratch);
    const unsigned qblocks = (unsigned)((nq + kQueriesPerBlock - 1) / kQueriesPerBlock);
    if (nchunks > 0) {
        dim3 grid(qblocks, (unsigned)nchunks, 1), block(kQueriesPerBlock, 1, 1);
        hipLaunchKernelGGL(knn2_kernel, grid, block, 0, s, reinterpret_cast<const uint4*>(q), nq,
                           reinterpret_cast<const uint4*>(db), ndb, rows, indexBase, part);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(knn2_fold_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, part,
                       (int)nchunks, nq, init, idx, d1, d2);
    return hipGetLastError();
}

hipError_t launch_knn2_merge(int nparts, int64_t nq, const int64_t* idx, const int32_t* d1,
                             const int32_t* d2, int32_t init, int64_t* oidx, int32_t* od1, int32_t* od2,
                             hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    hipLaunchKernelGGL(knn2_merge_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, nparts, nq,
                       idx, d1, d2, init, oidx, od1, od2);
    return hipGetLastError();
}

hipError_t launch_ratio_test(int64_t nq, const int32_t* d1, const int32_t* d2, int32_t th, int inclusive,
                             float ratio, uint8_t* accept, hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    hipLaunchKernelGGL(ratio_test_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, nq, d1, d2,
                       th, inclusive, ratio, accept);
    return hipGetLastError();
}

hipError_t launch_stereo_band(int npairs, int cap, const orbx_keypoint* kl, const uint8_t* dl,
                              const int32_t* nl, const orbx_keypoint* kr, const uint8_t* dr,
                              const int32_t* nr, const float* scale, int nlevels, float maxD,
                              int32_t* bestIdx, int32_t* bestDist, hipStream_t s) {
    if (npairs <= 0 || cap <= 0) return hipSuccess;
    dim3 grid((cap + 255) / 256, npairs, 1), block(256, 1, 1);
    hipLaunchKernelGGL(stereo_band_kernel, grid, block, 0, s, cap, kl, reinterpret_cast<const uint4*>(dl), nl,
                       kr, reinterpret_cast<const uint4*>(dr), nr, scale, nlevels, maxD, bestIdx, bestDist);
    return hipGetLastError();
}

}  // namespace orbx
