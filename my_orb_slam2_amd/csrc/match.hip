// 256-bit Hamming matching: ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1698-1714), the
// best / second-best scan that every Search* routine wraps around it (canonical form at
// src/ORBmatcher.cc:204-229) and the descriptor half of Frame::ComputeStereoMatches
// (src/Frame.cc:503-586).
//
// kNN kernel: the distances come off the matrix pipe. With a query's bits as s = +1 / -1 (bit 0 / 1) and a
// database row's bits as b = 0 / 1,
//     sum_k s[k] * b[k] = |row| - 2 <query, row> = D(query, row) - |query|,
// so a 32 x 32 tile of dot products over K = 256 ranks the 32 rows for each of 32 queries by Hamming distance,
// exactly (every product is 0 or +-1 and every sum a small integer). The bits are expanded to FP4 nibbles
// (1.0 = 0b0010, -1.0 = 0b1010) and a tile is four v_mfma_f32_32x32x64_f8f6f4 with both operands FP4: 128 matrix-pipe
// cycles for 1024 pairs, against 1072 VALU-issue cycles for the same pairs as xor + popcount.
// Queries are the N side and stay in registers as ready operands (NQB blocks of 32 per wave); database rows are the
// M side, 32 per step, expanded through a 256-entry table in LDS. A lane of the 32 x 32 result holds 16 rows of ONE
// query, so "can any of these rows enter this query's best two" is a three-level min tree and one compare, and the
// key updates are skipped wave-uniformly: after the first few thousand rows of a chunk that is almost always.
// The two best are kept as 32-bit keys (distance << 23 | row offset inside the chunk): taking min / median-of-3 on
// keys reproduces the reference's strict `<` updates, i.e. the lowest row index wins ties for the best and the
// second best is the second order statistic of the distances.
#include "orbx_internal.h"

#include <cmath>
#include <cstddef>
#include <mutex>

namespace orbx {

namespace {

constexpr int kWavesPerBlock = 4;            // independent waves; they share the expansion table only
constexpr int64_t kMaxChunkRows = 1 << 22;   // row offset must fit the 23-bit key field
constexpr unsigned kEmptyKey = 0xFFFFFFFFu;

struct Partial {       // result of one query over one database chunk
    int32_t d1, d2;    // distances, or INT32_MAX when the chunk had fewer than 1 / 2 rows
    int64_t idx;       // global row of d1, -1 if none
};

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Smallest of a tile's sixteen accumulators: seven v_min3_f32 and one v_min_f32, written out. fminf() would first
// pass an operand through v_max_f32(x, x) to quiet a signalling NaN, and the accumulators only ever hold small
// integers. The block begins with the wait states a vector instruction needs before it may read what a matrix
// instruction wrote (the assembler does not insert them for text it cannot see into; twelve is what the compiler
// places before its own first read of such a result).
__device__ __forceinline__ float min16f(const v16f& a) {
    float t0, t1, t2;
    asm("s_nop 11\n\t"
        "v_min3_f32 %0, %3, %4, %5\n\t"
        "v_min3_f32 %1, %6, %7, %8\n\t"
        "v_min3_f32 %2, %9, %10, %11\n\t"
        "v_min3_f32 %0, %0, %1, %12\n\t"
        "v_min3_f32 %1, %13, %14, %15\n\t"
        "v_min3_f32 %2, %2, %16, %17\n\t"
        "v_min3_f32 %0, %0, %1, %2\n\t"
        "v_min_f32 %0, %0, %18"
        : "=&v"(t0), "=&v"(t1), "=&v"(t2)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]),
          "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15]));
    return t0;
}

}  // namespace

// One wave owns 32 * NQB queries and walks the chunk's rows 32 at a time.
// Which descriptor bit meets which K position of the matrix instruction is free as long as both operands agree:
// lane (j, h) of a wave (j = lane % 32: the query of a block / the row of a step; h = lane / 32) supplies, for
// matrix instruction m, the 32 bits of descriptor dword 4h + m, eight bits to an operand dword through the table.
template <int NQB>
__global__ __launch_bounds__(64 * kWavesPerBlock, 2) void knn2_kernel(const uint32_t* __restrict__ query, int64_t nq,
                                                                       const uint32_t* __restrict__ db, int64_t ndb,
                                                                       int64_t chunkRows, int64_t indexBase,
                                                                       Partial* __restrict__ partial) {
    __shared__ uint32_t lut[256];                  // byte -> eight nibbles, 0b0010 (1.0) where the bit is set
    {
        const unsigned t = threadIdx.x;
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) v |= ((t >> i) & 1u) << (4 * i + 1);
        lut[t] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t qtile = ((int64_t)blockIdx.x * kWavesPerBlock + wv) * (32 * NQB);
    if (qtile >= nq) return;                       // no barrier below: waves are on their own from here

    // ---- this wave's queries as ready B operands: +1.0 = 0b0010, -1.0 = 0b1010 ----
    uint32_t B[NQB][4][4];
    int pq[NQB];                                   // |query|
    unsigned k1[NQB], k2[NQB];
    float thr[NQB];                                // (distance of k2) - |query|: what the accumulators are compared with
#pragma unroll
    for (int b = 0; b < NQB; b++) {
        const int64_t qi = min(qtile + 32 * b + j, nq - 1);
        const u32x4* qp = reinterpret_cast<const u32x4*>(query + qi * 8);
        const u32x4 lo = qp[0], hi = qp[1];
        pq[b] = __popc(lo.x) + __popc(lo.y) + __popc(lo.z) + __popc(lo.w) + __popc(hi.x) + __popc(hi.y) +
                __popc(hi.z) + __popc(hi.w);
        const u32x4 mine = h ? hi : lo;
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int k = 0; k < 4; k++) B[b][m][k] = 0x22222222u | (lut[(mine[m] >> (8 * k)) & 255u] << 2);
        k1[b] = k2[b] = kEmptyKey;
        thr[b] = (float)((int)(kEmptyKey >> 23) - pq[b]);     // 511 while there is no second best
    }

    // ---- the chunk's rows ----
    const int64_t j0 = (int64_t)blockIdx.y * chunkRows;
    const int rows = (int)min(chunkRows, ndb - j0);
    const int nsteps = (rows + 31) >> 5;
    // Rows come through a buffer resource over the chunk: a step's address is a scalar offset plus this lane's
    // constant, and the rows a last, partial step lacks read as zero (their keys are masked below).
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(db + j0 * 8), 0, rows * 32, 0x00027000);
    const int laneOff = j * 32 + 16 * h;           // the wave reads the step's 1024 bytes as one contiguous piece
    u32x4 raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, laneOff, 0, 0));
    for (int step = 0; step < nsteps; step++) {
        uint32_t A[4][4];
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int k = 0; k < 4; k++) A[m][k] = lut[(raw[m] >> (8 * k)) & 255u];
        if (step + 1 < nsteps)
            raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, laneOff, (step + 1) * 1024, 0));
        // accumulator r of this lane is row 8 * (r / 4) + 4 * h + r % 4 of the step
        const int rowBase = step * 32 + 4 * h;
#pragma unroll
        for (int b = 0; b < NQB; b++) {
            v16f acc = {0};
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const v8i av = {(int)A[m][0], (int)A[m][1], (int)A[m][2], (int)A[m][3], 0, 0, 0, 0};
                const v8i bv = {(int)B[b][m][0], (int)B[b][m][1], (int)B[b][m][2], (int)B[b][m][3], 0, 0, 0, 0};
                // format 4 = FP4 for both operands; scale operands that are the constant 0 select the instruction
                // without block scales (every block scale 1)
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 4, 4, 0, 0, 0, 0);
            }
            const float mn = min16f(acc);
            // A row can only change the best two if its distance is below the current second best (a lane meets its
            // rows in ascending order, so a later row carries a larger key than any key it holds).
            const unsigned long long hit = __builtin_amdgcn_ballot_w64(mn < thr[b]);
            if (hit == 0) continue;                                  // wave-uniform
            // key = (acc + |query|) << 23 | row
            const unsigned base = ((unsigned)pq[b] << 23) + (unsigned)rowBase;
            if (__popcll(hit) > 6) {               // many lanes have something to take (start of a chunk): no more tests
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int rr = 8 * (r >> 2) + (r & 3);
                    unsigned key = ((unsigned)(int)acc[r] << 23) + base + (unsigned)rr;
                    if (rowBase + rr >= rows) key = kEmptyKey;       // past the chunk's end (last step only)
                    k2[b] = max(k1[b], min(k2[b], key));             // the median of three, given k1 <= k2
                    k1[b] = min(k1[b], key);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    if (__builtin_amdgcn_ballot_w64(acc[r] < thr[b]) != 0) {
                        const int rr = 8 * (r >> 2) + (r & 3);
                        unsigned key = ((unsigned)(int)acc[r] << 23) + base + (unsigned)rr;
                        if (rowBase + rr >= rows) key = kEmptyKey;
                        k2[b] = max(k1[b], min(k2[b], key));
                        k1[b] = min(k1[b], key);
                    }
                }
            }
            thr[b] = (float)((int)(k2[b] >> 23) - pq[b]);
        }
    }
    // ---- lanes j and j + 32 hold the two halves of a query's rows: the best two of the union ----
#pragma unroll
    for (int b = 0; b < NQB; b++) {
        const unsigned o1 = (unsigned)__shfl_xor((int)k1[b], 32), o2 = (unsigned)__shfl_xor((int)k2[b], 32);
        const unsigned n1 = min(k1[b], o1), n2 = min(max(k1[b], o1), min(k2[b], o2));
        const int64_t qi = qtile + 32 * b + j;
        if (h == 0 && qi < nq) {
            Partial p;
            p.d1 = n1 == kEmptyKey ? INT32_MAX : (int32_t)(n1 >> 23);
            p.d2 = n2 == kEmptyKey ? INT32_MAX : (int32_t)(n2 >> 23);
            p.idx = n1 == kEmptyKey ? -1 : indexBase + j0 + (int64_t)(n1 & 0x7FFFFFu);
            partial[(int64_t)blockIdx.y * nq + qi] = p;
        }
    }
}

// Folds the per-chunk partials of one query in chunk (= index) order with the reference's update
// rule, then applies the start value `init`: a distance that is not < init never replaces it.
// Results go to three arrays, or (hits != nullptr) to one array of packed 16-byte records.
__global__ __launch_bounds__(256) void knn2_fold_kernel(const Partial* __restrict__ partial, int nchunks,
                                                        int64_t nq, int32_t init, int64_t* __restrict__ idx,
                                                        int32_t* __restrict__ d1, int32_t* __restrict__ d2,
                                                        Partial* __restrict__ hits) {
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
    if (hits) {
        Partial h;
        h.d1 = b1;
        h.d2 = b2;
        h.idx = bi;
        hits[qi] = h;
        return;
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

// The same fold over packed records: part p of query qi sits at hits[p * nq + qi], the layout one all-gather of every
// rank's nq records produces.
__global__ __launch_bounds__(256) void knn2_merge_packed_kernel(int nparts, int64_t nq,
                                                                const Partial* __restrict__ hits, int32_t init,
                                                                int64_t* __restrict__ oidx, int32_t* __restrict__ od1,
                                                                int32_t* __restrict__ od2) {
    const int64_t qi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    int32_t b1 = init, b2 = init;
    int64_t bi = -1;
    for (int p = 0; p < nparts; p++) {
        const Partial h = hits[(int64_t)p * nq + qi];
        if (h.idx >= 0 && h.d1 < b1) { b2 = b1; b1 = h.d1; bi = h.idx; }
        else if (h.idx >= 0 && h.d1 < b2) b2 = h.d1;
        if (h.d2 < b2) b2 = h.d2;
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

// Band search of Frame::ComputeStereoMatches (src/Frame.cc:513-586). The reference files every right
// keypoint under all image rows of its band (:520-530) and lets a left keypoint scan the list of its row.
// Here both sides of a pair are first counting-sorted by row in LDS (right keypoints by the first row of
// their band, left keypoints by their own row). A wave then holds 64 left keypoints of neighbouring rows
// and only walks the right keypoints whose band can reach those rows, a contiguous stretch of the sorted
// list found from the sort's row-start table. Right keypoints arrive through scalar loads; the 256-bit
// distance is computed only when some lane passes the row / octave / column tests. The best match is kept as
// the key (distance << 16 | right index), so the lowest index wins ties whatever the visiting order (:580
// is a strict `<` over ascending indices).
constexpr int kRowBins = 4096 + 64;     // images are at most 4096 rows; keys are clamped into [0, kRowBins)

struct StereoSorted {        // per pair, in the matcher's scratch
    int32_t* rStart;         // [kRowBins + 1] first sorted right keypoint whose band starts at row >= r
    uint32_t* rBand;         // [cap] (minr & 0xffff) | maxr << 16, sorted by max(minr, 0)
    float* rX;               // [cap] column
    uint32_t* rOctIdx;       // [cap] octave << 16 | index in the caller's array
    int32_t* lPerm;          // [cap] left keypoint indices sorted by row
    int32_t* matchCount;     // [1] left keypoints whose band match is close enough for the refinement ...
    int32_t* matchList;      // [cap] ... and their indices, in no particular order
};

__device__ __forceinline__ StereoSorted sorted_of_pair(uint8_t* work, int cap, int pair) {
    const size_t per = sizeof(int32_t) * ((size_t)kRowBins + 2 + 5 * (size_t)cap);
    uint8_t* p = work + per * pair;
    StereoSorted v;
    v.rStart = reinterpret_cast<int32_t*>(p);
    v.rBand = reinterpret_cast<uint32_t*>(v.rStart + kRowBins + 1);
    v.rX = reinterpret_cast<float*>(v.rBand + cap);
    v.rOctIdx = reinterpret_cast<uint32_t*>(v.rX + cap);
    v.lPerm = reinterpret_cast<int32_t*>(v.rOctIdx + cap);
    v.matchCount = v.lPerm + cap;
    v.matchList = v.matchCount + 1;
    return v;
}

// Exclusive scan of hist[0 .. kRowBins] (kRowBins + 1 entries, the last one receives the total). 256 threads.
__device__ void scan_rows(int* hist, int* part) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    constexpr int per = (kRowBins + 255) / 256;
    const int b = min(t * per, kRowBins), e = min(b + per, kRowBins);
    int s = 0;
    for (int i = b; i < e; i++) s += hist[i];
    int incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) part[wv] = incl;
    __syncthreads();
    int run = incl - s;
    for (int w = 0; w < wv; w++) run += part[w];
    for (int i = b; i < e; i++) {
        const int c = hist[i];
        hist[i] = run;
        run += c;
    }
    if (t == 255) hist[kRowBins] = run;
    __syncthreads();
}

// One workgroup per pair: counting sort of both sides by row.
__global__ __launch_bounds__(256) void stereo_sort_kernel(int cap, const orbx_keypoint* __restrict__ kl,
                                                          const int32_t* __restrict__ nl,
                                                          const orbx_keypoint* __restrict__ kr,
                                                          const int32_t* __restrict__ nr,
                                                          const float* __restrict__ scale, int nlevels,
                                                          uint8_t* __restrict__ work) {
    __shared__ int hist[kRowBins + 1];
    __shared__ int part[4];
    const int pair = blockIdx.x, t = threadIdx.x;
    const int nL = min(nl[pair], cap), nR = min(nr[pair], cap);
    const long long o = (long long)pair * cap;
    const StereoSorted S = sorted_of_pair(work, cap, pair);
    if (t == 0) *S.matchCount = 0;                 // the band search fills the list
    // ---- right side: key = first row of the band (:524-526) ----
    for (int i = t; i <= kRowBins; i += 256) hist[i] = 0;
    __syncthreads();
    for (int i = t; i < nR; i += 256) {
        const orbx_keypoint R = kr[o + i];
        const float r = __fmul_rn(2.0f, scale[min(max(R.octave, 0), nlevels - 1)]);
        const int minr = (int)floorf(__fsub_rn(R.y, r));
        atomicAdd(&hist[min(max(minr, 0), kRowBins - 1)], 1);
    }
    __syncthreads();
    scan_rows(hist, part);
    for (int i = t; i <= kRowBins; i += 256) S.rStart[i] = hist[i];
    __syncthreads();
    for (int i = t; i < nR; i += 256) {
        const orbx_keypoint R = kr[o + i];
        const float r = __fmul_rn(2.0f, scale[min(max(R.octave, 0), nlevels - 1)]);
        const int maxr = (int)ceilf(__fadd_rn(R.y, r)), minr = (int)floorf(__fsub_rn(R.y, r));
        const int pos = atomicAdd(&hist[min(max(minr, 0), kRowBins - 1)], 1);   // order inside a row is free
        S.rBand[pos] = ((uint32_t)minr & 0xffffu) | ((uint32_t)maxr << 16);
        S.rX[pos] = R.x;
        S.rOctIdx[pos] = ((uint32_t)R.octave << 16) | (uint32_t)i;
    }
    __syncthreads();
    // ---- left side: key = the row the keypoint looks up, (int)vL (:548) ----
    for (int i = t; i <= kRowBins; i += 256) hist[i] = 0;
    __syncthreads();
    for (int i = t; i < nL; i += 256) atomicAdd(&hist[min(max((int)kl[o + i].y, 0), kRowBins - 1)], 1);
    __syncthreads();
    scan_rows(hist, part);
    for (int i = t; i < nL; i += 256) S.lPerm[atomicAdd(&hist[min(max((int)kl[o + i].y, 0), kRowBins - 1)], 1)] = i;
}

constexpr int kStereoRefineDist = (100 + 50) / 2;   // thOrbDist = (TH_HIGH + TH_LOW) / 2 (src/Frame.cc:508)
constexpr int kRefineBlocks = 64;                   // workgroups per pair in the refinement

__global__ __launch_bounds__(256) void stereo_band_kernel(int cap, const orbx_keypoint* __restrict__ kl,
                                                          const uint4* __restrict__ dl, const int32_t* __restrict__ nl,
                                                          const uint4* __restrict__ dr, float maxD, int bandRows,
                                                          uint8_t* __restrict__ work, int32_t* __restrict__ bestIdx,
                                                          int32_t* __restrict__ bestDist) {
    // A workgroup owns 64 row-neighbouring left keypoints (one per lane, the same in each of its four waves); the
    // waves share the walk over the right keypoints, a quarter of every 64 each, and merge their bests through LDS.
    __shared__ unsigned merged[64];
    const int pair = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pos = blockIdx.x * 64 + lane;
    const int nL = min(nl[pair], cap);
    if (blockIdx.x * 64 >= nL) return;              // whole workgroup past the end (also nL == 0)
    if (wv == 0) merged[lane] = 100u << 16;         // TH_HIGH (:559), no index
    __syncthreads();
    const long long o = (long long)pair * cap;
    const StereoSorted S = sorted_of_pair(work, cap, pair);
    const bool live = pos < nL;
    const int iL = S.lPerm[live ? pos : nL - 1];    // idle lanes shadow the last keypoint
    const orbx_keypoint L = kl[o + iL];
    const uint4 qa = dl[2 * (o + iL)], qb = dl[2 * (o + iL) + 1];
    const int row = (int)L.y;                       // vRowIndices[vL], float -> index truncation (:548)
    const float minU = __fsub_rn(L.x, maxD), maxU = L.x;   // minD = 0 (:534,553-554)
    // the wave's rows are ascending; a band spans at most bandRows rows
    const int rowLo = __builtin_amdgcn_readfirstlane(row), rowHi = __builtin_amdgcn_readlane(row, 63);
    const int jBeg = S.rStart[min(max(rowLo - bandRows, 0), kRowBins)];
    const int jEnd = S.rStart[min(max(rowHi + 1, 0), kRowBins)];
    unsigned best = 100u << 16;                     // TH_HIGH (:559), no index
    // The stretch of right keypoints goes 64 at a time: every lane fetches one record and that keypoint's
    // descriptor (one round of memory latency per 64 candidates), then the wave walks the 64 with the fields
    // broadcast from the owning lane (v_readlane with a uniform index), so the tests run on scalars as before.
    for (int j0 = jBeg; j0 < jEnd; j0 += 64) {
        const int jm = min(j0 + lane, jEnd - 1);
        const uint32_t bandV = S.rBand[jm], oiV = S.rOctIdx[jm];
        const float rxV = S.rX[jm];
        const uint4 raV = dr[2 * (o + (oiV & 0xffffu))], rbV = dr[2 * (o + (oiV & 0xffffu)) + 1];
        const int n = min(16 * wv + 16, jEnd - j0);
        for (int u = 16 * wv; u < n; u++) {
            const uint32_t band = __builtin_amdgcn_readlane(bandV, u), oi = __builtin_amdgcn_readlane(oiV, u);
            const float rx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rxV), u));
            const int minr = (int)(short)(band & 0xffffu), maxr = (int)(short)(band >> 16);
            const int roct = (int)(short)(oi >> 16);
            const bool pass = maxU >= 0 && row >= minr && row <= maxr &&               // (:556, :548)
                              roct >= L.octave - 1 && roct <= L.octave + 1 &&          // (:570)
                              rx >= minU && rx <= maxU;                                // (:575)
            if (__builtin_amdgcn_ballot_w64(pass) == 0) continue;
            const uint32_t iR = oi & 0xffffu;
            const int d = __popc(qa.x ^ __builtin_amdgcn_readlane(raV.x, u)) + __popc(qa.y ^ __builtin_amdgcn_readlane(raV.y, u)) +
                          __popc(qa.z ^ __builtin_amdgcn_readlane(raV.z, u)) + __popc(qa.w ^ __builtin_amdgcn_readlane(raV.w, u)) +
                          __popc(qb.x ^ __builtin_amdgcn_readlane(rbV.x, u)) + __popc(qb.y ^ __builtin_amdgcn_readlane(rbV.y, u)) +
                          __popc(qb.z ^ __builtin_amdgcn_readlane(rbV.z, u)) + __popc(qb.w ^ __builtin_amdgcn_readlane(rbV.w, u));
            if (pass) best = min(best, ((unsigned)d << 16) | iR);   // d < 100 or nothing; lowest index wins ties (:580)
        }
    }
    atomicMin(&merged[lane], best);
    __syncthreads();
    if (live && wv == 0) {
        best = merged[lane];
        const int d = (int)(best >> 16);
        bestIdx[o + iL] = d < 100 ? (int)(best & 0xffffu) : -1;
        bestDist[o + iL] = d;
        if (d < kStereoRefineDist) S.matchList[atomicAdd(S.matchCount, 1)] = iL;   // (src/Frame.cc:589)
    }
}

// Sum over the wave; every lane must be active. The total is returned to all lanes.
__device__ __forceinline__ int wave_sum_i32(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false);  // row_half_mirror
    x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false);  // row_mirror
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast15
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);  // row_bcast31
    return __builtin_amdgcn_readlane(x, 63);
}

// Second half of Frame::ComputeStereoMatches (src/Frame.cc:589-661): for every left keypoint whose
// band match is closer than (TH_HIGH + TH_LOW) / 2, slide an 11x11 window +-5 px along the row of the
// right pyramid level, keep the offset with the smallest sum of absolute differences of the
// centre-subtracted patches, fit a parabola through its neighbours and turn the sub-pixel column into
// uRight and depth. One wave per listed left keypoint: a lane owns two of the 121 window positions, the eleven
// sums are wave reductions, everything after them is wave-uniform. sadDist receives the winning SAD (or -1).
__global__ __launch_bounds__(256) void stereo_refine_kernel(int cap, PyramidView left, PyramidView right,
                                                            const orbx_keypoint* __restrict__ kl,
                                                            const int32_t* __restrict__ nl,
                                                            const orbx_keypoint* __restrict__ kr,
                                                            const int32_t* __restrict__ bestIdx,
                                                            const int32_t* __restrict__ bestDist,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ invScale, float bf, float maxD,
                                                            float* __restrict__ uRight, float* __restrict__ depth,
                                                            int32_t* __restrict__ sadDist,
                                                            uint8_t* __restrict__ work) {
    const int pair = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long o = (long long)pair * cap;
    const int n = min(nl[pair], cap);
    const StereoSorted S = sorted_of_pair(work, cap, pair);
    // what is not refined has no match (its band match was too far, or there was none)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < cap; i += kRefineBlocks * 256)
        if (!(i < n && bestDist[o + i] < kStereoRefineDist)) {
            uRight[o + i] = -1.0f;
            depth[o + i] = -1.0f;
            sadDist[o + i] = -1;
        }
    // the left keypoints the band search listed, one per wave and turn
    const int count = *S.matchCount;
    for (int j = blockIdx.x * 4 + wv; j < count; j += kRefineBlocks * 4) {
        const int iL = S.matchList[j];
        float outU = -1.0f, outZ = -1.0f;
        int outD = -1;
        {
        const orbx_keypoint L = kl[o + iL];
        const int oct = min(max(L.octave, 0), left.nlevels - 1);
        const float uR0 = kr[o + bestIdx[o + iL]].x;
        const float sf = invScale[oct];
        const float scaleduL = roundf(__fmul_rn(L.x, sf)), scaledvL = roundf(__fmul_rn(L.y, sf));
        const float scaleduR0 = roundf(__fmul_rn(uR0, sf));
        const LevelView A = left.lv[oct], B = right.lv[oct];
        const int w = 5, Lr = 5;
        const float iniu = scaleduR0 + Lr - w, endu = scaleduR0 + Lr + w + 1;
        const int cu = (int)scaleduL, cv = (int)scaledvL, cr = (int)scaleduR0;
        // the reference indexes the pyramids unchecked; stay inside both levels
        const bool inside = cv - w >= 0 && cv + w < A.h && cv + w < B.h && cu - w >= 0 && cu + w < A.w &&
                            cr - Lr - w >= 0 && cr + Lr + w < B.w;
        if (!(iniu < 0 || endu >= (float)B.w) && inside) {     // (:612-613)
            const uint8_t* pa = A.base + (long long)pair * A.frameStride + (long long)cv * A.rowStride + cu;
            // window positions lane and lane + 64 (row-major 11 x 11; position 60 is the centre); lanes past
            // 120 repeat position 120 with weight 0
            const int i0 = lane, i1 = min(lane + 64, 120);
            const int dy0 = i0 / 11 - w, dx0 = i0 - (dy0 + w) * 11 - w;
            const int dy1 = i1 / 11 - w, dx1 = i1 - (dy1 + w) * 11 - w;
            const bool two = lane + 64 <= 120;
            const int l0 = pa[(long long)dy0 * A.rowStride + dx0], l1 = pa[(long long)dy1 * A.rowStride + dx1];
            // The eleven right-image bytes a position meets as the window slides are consecutive: three dwords
            // through a buffer resource over the level (byte offsets of any alignment; the twelfth byte may lie
            // one past the level's last row, where the resource returns zero instead of touching memory).
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<uint8_t*>(B.base + (long long)pair * B.frameStride), 0, (int)((long long)B.h * B.rowStride),
                0x00027000);
            const int off0 = (cv + dy0) * (int)B.rowStride + cr + dx0 - Lr, off1 = (cv + dy1) * (int)B.rowStride + cr + dx1 - Lr;
            uint32_t r0[3], r1[3];
#pragma unroll
            for (int q = 0; q < 3; q++) {
                r0[q] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off0 + 4 * q, 0, 0);
                r1[q] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off1 + 4 * q, 0, 0);
            }
            const int cL = __builtin_amdgcn_readlane(l0, 60);       // position 60 is the window's centre
            const int a0 = l0 - cL, a1 = l1 - cL;
            int dists[11];
            int best = INT32_MAX, bestInc = 0;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const int v0 = (int)((r0[k >> 2] >> (8 * (k & 3))) & 0xFFu), v1 = (int)((r1[k >> 2] >> (8 * (k & 3))) & 0xFFu);
                const int cR = __builtin_amdgcn_readlane(v0, 60);   // IR(cv, cr + k - Lr): the centre lane's own byte
                const int part = abs(a0 - (v0 - cR)) + (two ? abs(a1 - (v1 - cR)) : 0);
                const int sad = wave_sum_i32(part);
                dists[k] = sad;
                if (sad < best) { best = sad; bestInc = k - Lr; }           // float compare of exact integers (:622)
            }
            if (bestInc != -Lr && bestInc != Lr) {                      // (:631)
                float d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
                for (int i = 0; i < 11; i++) {                          // no dynamic register indexing
                    if (i == bestInc + Lr - 1) d1 = (float)dists[i];
                    if (i == bestInc + Lr) d2 = (float)dists[i];
                    if (i == bestInc + Lr + 1) d3 = (float)dists[i];
                }
                const float deltaR = __fdiv_rn(__fsub_rn(d1, d3),
                                               __fmul_rn(2.0f, __fsub_rn(__fadd_rn(d1, d3), __fmul_rn(2.0f, d2))));
                if (!(deltaR < -1 || deltaR > 1)) {                      // (:641)
                    float bestuR = __fmul_rn(scale[oct], __fadd_rn(__fadd_rn(scaleduR0, (float)bestInc), deltaR));
                    float disparity = __fsub_rn(L.x, bestuR);
                    if (disparity >= 0 && disparity < maxD) {            // (:649)
                        if (disparity <= 0) {
                            disparity = (float)0.01;
                            bestuR = (float)((double)L.x - 0.01);
                        }
                        outZ = __fdiv_rn(bf, disparity);
                        outU = bestuR;
                        outD = best;
                    }
                }
            }
        }
    }
        if (lane == 0) {
            uRight[o + iL] = outU;
            depth[o + iL] = outZ;
            sadDist[o + iL] = outD;
        }
    }
}

// Outlier cut of src/Frame.cc:663-676, one workgroup per pair: median SAD of the accepted matches (the
// element of rank n/2 in ascending order), threshold 1.5f * 1.4f * median, everything at or above it is
// reset to -1.
__global__ __launch_bounds__(256) void stereo_median_cut_kernel(int cap, const int32_t* __restrict__ nl,
                                                                const int32_t* __restrict__ sadDist,
                                                                float* __restrict__ uRight, float* __restrict__ depth) {
    __shared__ int s_cnt[256];
    const int pair = blockIdx.x, t = threadIdx.x;
    const long long o = (long long)pair * cap;
    const int n = min(nl[pair], cap);
    auto count_le = [&](int v, int& total) {      // #accepted with dist <= v, and #accepted
        int c = 0, a = 0;
        for (int i = t; i < n; i += 256) {
            const int d = sadDist[o + i];
            a += d >= 0;
            c += d >= 0 && d <= v;
        }
        s_cnt[t] = c;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (t < off) s_cnt[t] += s_cnt[t + off];
            __syncthreads();
        }
        c = s_cnt[0];
        __syncthreads();
        s_cnt[t] = a;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (t < off) s_cnt[t] += s_cnt[t + off];
            __syncthreads();
        }
        total = s_cnt[0];
        __syncthreads();
        return c;
    };
    int lo = 0, hi = 121 * 510;                   // largest possible 11x11 SAD of centred bytes
    constexpr int kKeep = 8;                      // values a thread keeps in registers (pairs of up to 2048 keypoints)
    if (n <= 256 * kKeep) {
        // the bisection on values read once, with wave sums and one barrier per step
        __shared__ int s_part[2][4];
        const int lane = t & 63, wv = t >> 6;
        int v[kKeep];
#pragma unroll
        for (int k = 0; k < kKeep; k++) v[k] = t + 256 * k < n ? sadDist[o + t + 256 * k] : -1;
        int step = 0;
        auto block_count_le = [&](int limit) {    // #accepted with dist <= limit
            int c = 0;
#pragma unroll
            for (int k = 0; k < kKeep; k++) c += v[k] >= 0 && v[k] <= limit;
            c = wave_sum_i32(c);
            if (lane == 0) s_part[step & 1][wv] = c;
            __syncthreads();
            const int* p = s_part[step & 1];
            step++;
            return p[0] + p[1] + p[2] + p[3];
        };
        const int total = block_count_le(hi);
        if (total == 0) return;                   // the reference would index an empty vector here
        const int want = total / 2 + 1;           // smallest v with #(d <= v) >= rank + 1
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (block_count_le(mid) >= want) hi = mid; else lo = mid + 1;
        }
    } else {
        int total = 0;
        count_le(0, total);
        if (total == 0) return;
        const int want = total / 2 + 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int dummy;
            if (count_le(mid, dummy) >= want) hi = mid; else lo = mid + 1;
        }
    }
    const float thDist = __fmul_rn(__fmul_rn(1.5f, 1.4f), (float)lo);
    for (int i = t; i < n; i += 256) {
        const int d = sadDist[o + i];
        if (d >= 0 && !((float)d < thDist)) {
            uRight[o + i] = -1.0f;
            depth[o + i] = -1.0f;
        }
    }
}

namespace {
// How a scan is cut up: query blocks per wave (8, 2 or 1) and rows per chunk.
// A wave that holds more queries spends less on expanding rows (one expansion serves 32 * nqb queries), but the scan
// needs about two waves per SIMD to keep the matrix pipe fed, and every chunk starts with a few thousand rows that
// mostly take the update path, so chunks should be long. Largest nqb first; the first that fills the chip wins.
struct ScanPlan {
    int nqb;
    int64_t rows, nchunks;
};
ScanPlan plan_scan(int64_t nq, int64_t ndb) {
    const int64_t kWavesWanted = 2048, kWavesEnough = 1024;
    const int nqbs[3] = {8, 2, 1};
    const int64_t minRows[3] = {65536, 8192, 1024};
    ScanPlan plan{1, 1, 0};
    for (int i = 0; i < 3; i++) {
        const int64_t qwaves = (nq + 32 * nqbs[i] - 1) / (32 * nqbs[i]);
        int64_t want = (kWavesWanted + qwaves - 1) / qwaves;                // chunks wanted
        int64_t rows = ndb > 0 ? (ndb + want - 1) / want : 1;
        rows = std::min(std::max(rows, minRows[i]), kMaxChunkRows);
        // equal chunks: a short last chunk would leave most of the chip idle while the long ones finish
        int64_t nchunks = ndb > 0 ? (ndb + rows - 1) / rows : 0;
        if (ndb > 0) rows = (ndb + nchunks - 1) / nchunks;
        plan = ScanPlan{nqbs[i], rows, nchunks};
        if (qwaves * nchunks >= kWavesEnough) break;
    }
    return plan;
}
}  // namespace

size_t knn2_scratch_bytes(int64_t nq, int64_t ndb) {
    if (nq <= 0) return 0;
    const ScanPlan plan = plan_scan(nq, ndb);
    return (size_t)std::max<int64_t>(plan.nchunks, 1) * (size_t)nq * sizeof(Partial);
}

namespace {
// The scan over chunks of `rows` database rows into `part`, then the fold in chunk order.
hipError_t launch_knn2_chunks(const uint8_t* q, int64_t nq, const uint8_t* db, int64_t ndb, const ScanPlan& plan,
                              int32_t init, int64_t indexBase, int64_t* idx, int32_t* d1, int32_t* d2, void* hits,
                              Partial* part, hipStream_t s) {
    if (plan.nchunks > 65535 || plan.rows > kMaxChunkRows) return hipErrorInvalidValue;
    if (plan.nchunks > 0) {
        const int perBlock = 32 * plan.nqb * kWavesPerBlock;
        dim3 grid((unsigned)((nq + perBlock - 1) / perBlock), (unsigned)plan.nchunks, 1), block(64 * kWavesPerBlock, 1, 1);
        const uint32_t* q32 = reinterpret_cast<const uint32_t*>(q);
        const uint32_t* db32 = reinterpret_cast<const uint32_t*>(db);
        if (plan.nqb == 8)
            hipLaunchKernelGGL(knn2_kernel<8>, grid, block, 0, s, q32, nq, db32, ndb, plan.rows, indexBase, part);
        else if (plan.nqb == 2)
            hipLaunchKernelGGL(knn2_kernel<2>, grid, block, 0, s, q32, nq, db32, ndb, plan.rows, indexBase, part);
        else
            hipLaunchKernelGGL(knn2_kernel<1>, grid, block, 0, s, q32, nq, db32, ndb, plan.rows, indexBase, part);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(knn2_fold_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, part,
                       (int)plan.nchunks, nq, init, idx, d1, d2, static_cast<Partial*>(hits));
    return hipGetLastError();
}
}  // namespace

static_assert(sizeof(Partial) == sizeof(orbx_knn2_hit) && offsetof(Partial, idx) == offsetof(orbx_knn2_hit, idx) &&
                  offsetof(Partial, d2) == offsetof(orbx_knn2_hit, d2),
              "the per-chunk partial result and the packed result record share one layout");

hipError_t launch_knn2(const uint8_t* q, int64_t nq, const uint8_t* db, int64_t ndb, int32_t init,
                       int64_t indexBase, int64_t* idx, int32_t* d1, int32_t* d2, void* hits, void* scratch,
                       size_t scratchBytes, hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    const ScanPlan plan = plan_scan(nq, ndb);
    if ((size_t)plan.nchunks * (size_t)nq * sizeof(Partial) > scratchBytes) return hipErrorInvalidValue;
    return launch_knn2_chunks(q, nq, db, ndb, plan, init, indexBase, idx, d1, d2, hits,
                              static_cast<Partial*>(scratch), s);
}

// ---- the same scan for callers that cannot allocate ---------------------------------------------
// The per-chunk partial results live in a fixed array inside the library's code object instead of a buffer
// from hipMalloc. Every GPU the library runs on holds its own copy of the array, at its own address, until the
// process ends. In host code the array's name is only a handle for hipGetSymbolAddress; the address a kernel
// can use is looked up once per device and kept in g_staticPartial.
constexpr size_t kStaticPartialBytes = (size_t)32 << 20;
constexpr int64_t kStaticPartials = (int64_t)(kStaticPartialBytes / sizeof(Partial));   // 2 Mi (query, chunk) pairs
constexpr int kMaxStaticDevices = 64;
static __device__ __align__(256) unsigned char scratch[kStaticPartialBytes];

namespace {
std::mutex g_staticMutex;
Partial* g_staticPartial[kMaxStaticDevices] = {};

// The scan's usual plan, if its nq * nchunks partials fit the fixed array.
bool plan_scan_static(int64_t nq, int64_t ndb, ScanPlan* out) {
    const ScanPlan plan = plan_scan(nq, ndb);
    if (nq > kStaticPartials || plan.nchunks * nq > kStaticPartials) return false;
    *out = plan;
    return true;
}
}  // namespace

int64_t knn2_static_max_partials() { return kStaticPartials; }

bool knn2_static_fits(int64_t nq, int64_t ndb) {
    ScanPlan plan;
    return plan_scan_static(nq, ndb, &plan);
}

// Looks up this device's copy of the array. `device` must be the current device.
hipError_t knn2_static_init(int device) {
    if (device < 0 || device >= kMaxStaticDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_staticMutex);
    if (g_staticPartial[device]) return hipSuccess;
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    g_staticPartial[device] = static_cast<Partial*>(p);
    return hipSuccess;
}

// `device` must be the current device and knn2_static_fits(nq, ndb) must hold. Every partial the fold reads is
// written by the scan of the same call, so nothing in the array has to be cleared between calls.
hipError_t launch_knn2_static(int device, const uint8_t* q, int64_t nq, const uint8_t* db, int64_t ndb,
                              int32_t init, int64_t indexBase, int64_t* idx, int32_t* d1, int32_t* d2,
                              hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    ScanPlan plan;
    if (!plan_scan_static(nq, ndb, &plan)) return hipErrorInvalidValue;
    hipError_t e = knn2_static_init(device);
    if (e != hipSuccess) return e;
    Partial* part;
    {
        std::lock_guard<std::mutex> lock(g_staticMutex);
        part = g_staticPartial[device];
    }
    return launch_knn2_chunks(q, nq, db, ndb, plan, init, indexBase, idx, d1, d2, nullptr, part, s);
}

hipError_t launch_knn2_merge(int nparts, int64_t nq, const int64_t* idx, const int32_t* d1,
                             const int32_t* d2, int32_t init, int64_t* oidx, int32_t* od1, int32_t* od2,
                             hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    hipLaunchKernelGGL(knn2_merge_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, nparts, nq,
                       idx, d1, d2, init, oidx, od1, od2);
    return hipGetLastError();
}

hipError_t launch_knn2_merge_packed(int nparts, int64_t nq, const void* hits, int32_t init, int64_t* oidx,
                                    int32_t* od1, int32_t* od2, hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    hipLaunchKernelGGL(knn2_merge_packed_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, nparts, nq,
                       static_cast<const Partial*>(hits), init, oidx, od1, od2);
    return hipGetLastError();
}

hipError_t launch_ratio_test(int64_t nq, const int32_t* d1, const int32_t* d2, int32_t th, int inclusive,
                             float ratio, uint8_t* accept, hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    hipLaunchKernelGGL(ratio_test_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, nq, d1, d2,
                       th, inclusive, ratio, accept);
    return hipGetLastError();
}

size_t stereo_band_scratch_bytes(int npairs, int cap) {
    return sizeof(int32_t) * ((size_t)kRowBins + 2 + 5 * (size_t)cap) * (size_t)npairs;
}

// maxScale: the largest scale factor (a band is 2 * scale rows either side of the keypoint).
hipError_t launch_stereo_band(int npairs, int cap, const orbx_keypoint* kl, const uint8_t* dl,
                              const int32_t* nl, const orbx_keypoint* kr, const uint8_t* dr,
                              const int32_t* nr, const float* scale, int nlevels, float maxScale, float maxD,
                              void* work, int32_t* bestIdx, int32_t* bestDist, hipStream_t s) {
    if (npairs <= 0 || cap <= 0) return hipSuccess;
    if (cap > 65535) return hipErrorInvalidValue;                 // the right index shares a key with the distance
    hipLaunchKernelGGL(stereo_sort_kernel, dim3(npairs), dim3(256), 0, s, cap, kl, nl, kr, nr, scale, nlevels,
                       static_cast<uint8_t*>(work));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int bandRows = 2 * (int)std::ceil(2.0f * maxScale) + 2;   // maxr - minr never exceeds this
    dim3 grid((cap + 63) / 64, npairs, 1), block(256, 1, 1);
    hipLaunchKernelGGL(stereo_band_kernel, grid, block, 0, s, cap, kl, reinterpret_cast<const uint4*>(dl), nl,
                       reinterpret_cast<const uint4*>(dr), maxD, bandRows, static_cast<uint8_t*>(work), bestIdx,
                       bestDist);
    return hipGetLastError();
}

hipError_t launch_stereo_refine(int npairs, int cap, const PyramidView& left, const PyramidView& right,
                                const orbx_keypoint* kl, const int32_t* nl, const orbx_keypoint* kr,
                                const int32_t* bestIdx, const int32_t* bestDist, const float* scale,
                                const float* invScale, float bf, float maxD, float* uRight, float* depth,
                                int32_t* sadDist, void* work, hipStream_t s) {
    if (npairs <= 0 || cap <= 0) return hipSuccess;
    hipLaunchKernelGGL(stereo_refine_kernel, dim3(kRefineBlocks, npairs, 1), dim3(256), 0, s, cap, left, right,
                       kl, nl, kr, bestIdx, bestDist, scale, invScale, bf, maxD, uRight, depth, sadDist,
                       static_cast<uint8_t*>(work));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(stereo_median_cut_kernel, dim3(npairs), dim3(256), 0, s, cap, nl, sadDist, uRight, depth);
    return hipGetLastError();
}

}  // namespace orbx
