// Experiment: the best / second-best Hamming scan (src/ORBmatcher.cc:204-229 around :1698-1714) on the matrix pipe.
// With query bits as s = +1 / -1 (bit 0 / 1) and database bits as b = 0 / 1,
//     sum_k s_q[k] * b_r[k] = |r| - 2 <q, r> = D(q, r) - |q|,
// so a 32 x 32 tile of dot products over K = 256 ranks the rows of every query by distance, exactly (all values are
// small integers). MODE 0: v_mfma_i32_32x32x32_i8 (bits expanded to bytes); MODE 1: v_mfma_scale_f32_32x32x64_f8f6f4 with
// FP4 operands (bits expanded to nibbles, 1.0 = 0b0010, -1.0 = 0b1010, block scales 2^0).
// Queries are the N side and stay in registers (NQB blocks of 32 per wave); database rows are the M side, 32 per
// step. A lane of the 32 x 32 result holds 16 rows of ONE query, so "can any of these rows change the best two" is a
// min tree and one compare against the query's threshold, and the key updates are skipped wave-uniformly.
//   usage: knn_mfma_probe <mode> <nq> <ndb> <chunk_rows> [check] [expand: 0 LUT in LDS, 1 v_perm, 2 LUT copy per bank]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Partial { int32_t d1, d2; int64_t idx; };
constexpr unsigned kEmptyKey = 0xFFFFFFFFu;

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c) { return max(min(a, b), min(max(a, b), c)); }

template <int MODE, int NQB, int EXP>
__global__ __launch_bounds__(256, 2) void knn2_mfma_kernel(const uint32_t* __restrict__ query, int64_t nq,
                                                           const uint32_t* __restrict__ db, int64_t ndb,
                                                           int64_t chunkRows, int64_t indexBase,
                                                           Partial* __restrict__ partial) {
    // MODE 0: byte -> 8 bytes of 0/1 (uint2); MODE 1: byte -> 8 nibbles of 0b0000 / 0b0010 (uint32)
    __shared__ __attribute__((aligned(16))) uint32_t lut[MODE == 0 ? 512 : 256];
    __shared__ uint32_t lutRep[EXP == 2 ? 256 * 32 : 1];
    {
        const unsigned t = threadIdx.x;
        if (MODE == 0) {
            uint32_t lo = 0, hi = 0;
            for (int i = 0; i < 4; i++) {
                lo |= ((t >> i) & 1u) << (8 * i);
                hi |= ((t >> (4 + i)) & 1u) << (8 * i);
            }
            lut[2 * t] = lo;
            lut[2 * t + 1] = hi;
        } else {
            uint32_t v = 0;
            for (int i = 0; i < 8; i++) v |= ((t >> i) & 1u) << (4 * i + 1);
            lut[t] = v;
        }
    }
    __syncthreads();
    if (EXP == 2) {
        for (int i = threadIdx.x; i < 256 * 32; i += 256) lutRep[i] = lut[i >> 5];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    constexpr int NM = MODE == 0 ? 8 : 4;          // matrix instructions per 32 x 32 tile (K = 256)
    const int64_t qtile = ((int64_t)blockIdx.x * 4 + wv) * (32 * NQB);
    if (qtile >= nq) return;

    // ---- this wave's queries as the B operands ----
    uint32_t B[NQB][NM][4];
    int pq[NQB];
#pragma unroll
    for (int b = 0; b < NQB; b++) {
        const int64_t qi = min(qtile + 32 * b + j, nq - 1);
        const uint32_t* qp = query + qi * 8;
        int pc = 0;
#pragma unroll
        for (int d = 0; d < 8; d++) pc += __popc(qp[d]);
        pq[b] = pc;
        if (MODE == 0) {
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const uint32_t w = qp[c] >> (16 * h);
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const uint32_t byte = (w >> (8 * k)) & 255u;
                    // 0 -> +1 (0x01), 1 -> -1 (0xFF)
                    B[b][c][2 * k] = (lut[2 * byte] * 0xFFu) | 0x01010101u;
                    B[b][c][2 * k + 1] = (lut[2 * byte + 1] * 0xFFu) | 0x01010101u;
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t w = qp[2 * m + h];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (EXP == 1) B[b][m][k] = __builtin_amdgcn_perm(0u, 0xAAA22A22u, (w >> (2 * k)) & 0x03030303u);
                    else B[b][m][k] = 0x22222222u | (lut[(w >> (8 * k)) & 255u] << 2);
                }
            }
        }
    }
    unsigned k1[NQB], k2[NQB];
    int thr[NQB];                                   // (distance of k2) - |q|
#pragma unroll
    for (int b = 0; b < NQB; b++) {
        k1[b] = k2[b] = kEmptyKey;
        thr[b] = (int)(kEmptyKey >> 23) - pq[b];
    }

    // ---- the chunk's rows, 32 per step ----
    const int64_t j0 = (int64_t)blockIdx.y * chunkRows;
    const int rows = (int)min(chunkRows, ndb - j0);
    const int nrb = (rows + 31) >> 5;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(db + j0 * 8), 0, rows * 32, 0x00027000);
    // lane (row j of the step, half h) needs 16 bytes of its row: MODE 0 the h-th 16-bit half of every dword (it
    // loads the whole row), MODE 1 dwords h, 2 + h, 4 + h, 6 + h
    const int laneOff = j * 32 + (MODE == 0 ? 0 : 4 * h);
    uint32_t raw[MODE == 0 ? 8 : 4];
    auto load_rows = [&](int rb) {
        if (MODE == 0) {
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            const u4 a = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, laneOff, rb * 1024, 0));
            const u4 c = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, laneOff + 16, rb * 1024, 0));
            raw[0] = a.x; raw[1] = a.y; raw[2] = a.z; raw[3] = a.w;
            raw[4] = c.x; raw[5] = c.y; raw[6] = c.z; raw[7] = c.w;
        } else {
#pragma unroll
            for (int m = 0; m < 4; m++) raw[m] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, laneOff + 8 * m, rb * 1024, 0);
        }
    };
    load_rows(0);
    for (int rb = 0; rb < nrb; rb++) {
        uint32_t A[NM][4];
        if (MODE == 0) {
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const uint32_t w = raw[c] >> (16 * h);
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const uint2 e = *reinterpret_cast<const uint2*>(&lut[2 * ((w >> (8 * k)) & 255u)]);
                    A[c][2 * k] = e.x;
                    A[c][2 * k + 1] = e.y;
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t byte = (raw[m] >> (8 * k)) & 255u;
                    if (EXP == 0) A[m][k] = lut[byte];
                    else if (EXP == 2) A[m][k] = lutRep[byte * 32 + (lane & 31)];
                    else {
                        // Which descriptor bit lands on which K position is free as long as both operands agree:
                        // here operand dword k takes bits 2k, 2k+1 of every byte of the descriptor dword, so its
                        // v_perm_b32 selector is one shift and one mask; the four byte patterns 00, 02, 20, 22
                        // hold the nibble pairs of two bits.
                        const uint32_t sel = (raw[m] >> (2 * k)) & 0x03030303u;
                        A[m][k] = __builtin_amdgcn_perm(0u, 0x22200200u, sel);
                    }
                }
        }
        if (rb + 1 < nrb) load_rows(rb + 1);
        const int rowBase = rb * 32 + 4 * h;
#pragma unroll
        for (int b = 0; b < NQB; b++) {
            int accI[16];
            if (MODE == 0) {
                v16i acc = {0};
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const v4i a = {(int)A[c][0], (int)A[c][1], (int)A[c][2], (int)A[c][3]};
                    const v4i bb = {(int)B[b][c][0], (int)B[b][c][1], (int)B[b][c][2], (int)B[b][c][3]};
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bb, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; r++) accI[r] = acc[r];
            } else {
                v16f acc = {0};
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const v8i a = {(int)A[m][0], (int)A[m][1], (int)A[m][2], (int)A[m][3], 0, 0, 0, 0};
                    const v8i bb = {(int)B[b][m][0], (int)B[b][m][1], (int)B[b][m][2], (int)B[b][m][3], 0, 0, 0, 0};
                    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bb, acc, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                }
                // exact small integers; compared as floats, converted only on the update path
#pragma unroll
                for (int r = 0; r < 16; r++) accI[r] = __float_as_int(acc[r]);
            }
            bool any;
            if (MODE == 0) {
                const int m0 = min(min(accI[0], accI[1]), accI[2]), m1 = min(min(accI[3], accI[4]), accI[5]);
                const int m2 = min(min(accI[6], accI[7]), accI[8]), m3 = min(min(accI[9], accI[10]), accI[11]);
                const int m4 = min(min(accI[12], accI[13]), accI[14]);
                const int m = min(min(min(m0, m1), m2), min(min(m3, m4), accI[15]));
                any = __builtin_amdgcn_ballot_w64(m < thr[b]) != 0;
            } else {
                auto f = [&](int r) { return __int_as_float(accI[r]); };
                const float m0 = fminf(fminf(f(0), f(1)), f(2)), m1 = fminf(fminf(f(3), f(4)), f(5));
                const float m2 = fminf(fminf(f(6), f(7)), f(8)), m3 = fminf(fminf(f(9), f(10)), f(11));
                const float m4 = fminf(fminf(f(12), f(13)), f(14));
                const float m = fminf(fminf(fminf(m0, m1), m2), fminf(fminf(m3, m4), f(15)));
                any = __builtin_amdgcn_ballot_w64(m < (float)thr[b]) != 0;
            }
            if (any) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int a = MODE == 0 ? accI[r] : (int)__int_as_float(accI[r]);
                    if (__builtin_amdgcn_ballot_w64(a < thr[b]) != 0) {
                        const int row = rowBase + 8 * (r >> 2) + (r & 3);
                        const unsigned key = row < rows ? ((unsigned)(a + pq[b]) << 23) | (unsigned)row : kEmptyKey;
                        k2[b] = umed3(k1[b], k2[b], key);
                        k1[b] = min(k1[b], key);
                    }
                }
                thr[b] = (int)(k2[b] >> 23) - pq[b];
            }
        }
    }
    // ---- the two halves of a query's rows sit in lanes j and j + 32 ----
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


// ---- second form: the candidate for the product ----
// FP4 only; a lane reads 16 contiguous bytes of its row (half h of the row: dwords 4h .. 4h+3; matrix instruction m
// takes dword 4h + m from both operands), tiles go in pairs on two accumulator sets when PAIR is set, and the update path is
// branch-free when many lanes have a row to take and filtered register by register otherwise.
template <int NQB, bool PAIR, int WPS, bool UNSCALED = false, int DIAG = 0>
__global__ __launch_bounds__(256, WPS) void knn2_fp4_kernel(const uint32_t* __restrict__ query, int64_t nq,
                                                            const uint32_t* __restrict__ db, int64_t ndb,
                                                            int64_t chunkRows, int64_t indexBase,
                                                            Partial* __restrict__ partial) {
    __shared__ uint32_t lut[256];
    {
        const unsigned t = threadIdx.x;
        uint32_t v = 0;
        for (int i = 0; i < 8; i++) v |= ((t >> i) & 1u) << (4 * i + 1);
        lut[t] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t qtile = ((int64_t)blockIdx.x * 4 + wv) * (32 * NQB);
    if (qtile >= nq) return;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    uint32_t B[NQB][4][4];
    int pq[NQB];
#pragma unroll
    for (int b = 0; b < NQB; b++) {
        const int64_t qi = min(qtile + 32 * b + j, nq - 1);
        const u4* qp = reinterpret_cast<const u4*>(query + qi * 8);
        const u4 lo = qp[0], hi = qp[1];
        pq[b] = __popc(lo.x) + __popc(lo.y) + __popc(lo.z) + __popc(lo.w) + __popc(hi.x) + __popc(hi.y) + __popc(hi.z) + __popc(hi.w);
        const u4 mine = h ? hi : lo;
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int k = 0; k < 4; k++) B[b][m][k] = 0x22222222u | (lut[(mine[m] >> (8 * k)) & 255u] << 2);
    }
    unsigned k1[NQB], k2[NQB];
    float thr[NQB];                                 // (distance of k2) - |q|, as the accumulators hold it
#pragma unroll
    for (int b = 0; b < NQB; b++) {
        k1[b] = k2[b] = kEmptyKey;
        thr[b] = (float)((int)(kEmptyKey >> 23) - pq[b]);
    }
    const int64_t j0 = (int64_t)blockIdx.y * chunkRows;
    const int rows = (int)min(chunkRows, ndb - j0);
    const int nrb = (rows + 31) >> 5;
    // DIAG == 3: `db` holds the rows already expanded to FP4 (128 bytes per row, dword d of a row as 16 bytes at
    // d * 16): no table look-ups in the loop, four times the bytes
    constexpr bool PREX = DIAG == 3;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(db + j0 * (PREX ? 32 : 8)), 0, rows * (PREX ? 128 : 32), 0x00027000);
    const int laneOff = PREX ? j * 128 + 64 * h : j * 32 + 16 * h;
    u4 raw = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, laneOff, 0, 0));
    u4 rawx[3];
    if (PREX) {
#pragma unroll
        for (int m = 1; m < 4; m++) rawx[m - 1] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, laneOff + 16 * m, 0, 0));
    }

    auto tile = [&](const uint32_t (&A)[4][4], int b) {
        v16f acc = {0};
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const v8i a = {(int)A[m][0], (int)A[m][1], (int)A[m][2], (int)A[m][3], 0, 0, 0, 0};
            const v8i bb = {(int)B[b][m][0], (int)B[b][m][1], (int)B[b][m][2], (int)B[b][m][3], 0, 0, 0, 0};
            // scale operands that are the constant 0 select the instruction without block scales (scale 1)
            if (UNSCALED) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bb, acc, 4, 4, 0, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bb, acc, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
        return acc;
    };
    auto epilogue = [&](const v16f& acc, int b, int rowBase) {
        const float m0 = fminf(fminf(acc[0], acc[1]), acc[2]), m1 = fminf(fminf(acc[3], acc[4]), acc[5]);
        const float m2 = fminf(fminf(acc[6], acc[7]), acc[8]), m3 = fminf(fminf(acc[9], acc[10]), acc[11]);
        const float m4 = fminf(fminf(acc[12], acc[13]), acc[14]);
        const float mn = fminf(fminf(fminf(m0, m1), m2), fminf(fminf(m3, m4), acc[15]));
        const unsigned long long hit = __builtin_amdgcn_ballot_w64(mn < thr[b]);
        if (hit == 0) return;
        if (DIAG == 1) { k1[b] = min(k1[b], (unsigned)hit); return; }     // ceiling probe: no update path (wrong results)
        // key = (acc + |q|) << 23 | row; rows past the chunk's end (last step only) never enter
        const unsigned base = ((unsigned)pq[b] << 23) + (unsigned)rowBase;
        if (__popcll(hit) > 6) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = rowBase + 8 * (r >> 2) + (r & 3);
                unsigned key = ((unsigned)(int)acc[r] << 23) + base + (unsigned)(8 * (r >> 2) + (r & 3));
                if (row >= rows) key = kEmptyKey;
                k2[b] = max(k1[b], min(k2[b], key));     // the median of three, given k1 <= k2
                k1[b] = min(k1[b], key);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                if (__builtin_amdgcn_ballot_w64(acc[r] < thr[b]) != 0) {
                    const int row = rowBase + 8 * (r >> 2) + (r & 3);
                    unsigned key = ((unsigned)(int)acc[r] << 23) + base + (unsigned)(8 * (r >> 2) + (r & 3));
                    if (row >= rows) key = kEmptyKey;
                    k2[b] = max(k1[b], min(k2[b], key));
                    k1[b] = min(k1[b], key);
                }
            }
        }
        thr[b] = (float)((int)(k2[b] >> 23) - pq[b]);
    };

    for (int rb = 0; rb < nrb; rb++) {
        uint32_t A[4][4];
        if (PREX) {
#pragma unroll
            for (int k = 0; k < 4; k++) A[0][k] = raw[k];
#pragma unroll
            for (int m = 1; m < 4; m++)
#pragma unroll
                for (int k = 0; k < 4; k++) A[m][k] = rawx[m - 1][k];
            if (rb + 1 < nrb) {
                raw = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, laneOff, (rb + 1) * 4096, 0));
#pragma unroll
                for (int m = 1; m < 4; m++)
                    rawx[m - 1] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, laneOff + 16 * m, (rb + 1) * 4096, 0));
            }
        } else {
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int k = 0; k < 4; k++) A[m][k] = lut[(raw[m] >> (8 * k)) & 255u];
            if (rb + 1 < nrb) raw = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, laneOff, (rb + 1) * 1024, 0));
        }
        const int rowBase = rb * 32 + 4 * h;
        if (PAIR && DIAG == 2) {
            // both tiles' matrix instructions, both min trees, ONE branch for the pair
#pragma unroll
            for (int b = 0; b < NQB; b += 2) {
                const v16f x = tile(A, b);
                const v16f y = tile(A, b + 1);
                auto tmin = [&](const v16f& acc) {
                    const float m0 = fminf(fminf(acc[0], acc[1]), acc[2]), m1 = fminf(fminf(acc[3], acc[4]), acc[5]);
                    const float m2 = fminf(fminf(acc[6], acc[7]), acc[8]), m3 = fminf(fminf(acc[9], acc[10]), acc[11]);
                    const float m4 = fminf(fminf(acc[12], acc[13]), acc[14]);
                    return fminf(fminf(fminf(m0, m1), m2), fminf(fminf(m3, m4), acc[15]));
                };
                const float mx = tmin(x) - thr[b], my = tmin(y) - thr[b + 1];
                if (__builtin_amdgcn_ballot_w64(fminf(mx, my) < 0.0f) != 0) {
                    epilogue(x, b, rowBase);
                    epilogue(y, b + 1, rowBase);
                }
            }
        } else if (PAIR) {
#pragma unroll
            for (int b = 0; b < NQB; b += 2) {
                const v16f x = tile(A, b);
                const v16f y = tile(A, b + 1);
                epilogue(x, b, rowBase);
                epilogue(y, b + 1, rowBase);
            }
        } else {
#pragma unroll
            for (int b = 0; b < NQB; b++) epilogue(tile(A, b), b, rowBase);
        }
    }
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


// ---- third form: no branch between the tiles of a step ----
// The eight tiles of a step are issued as one straight block (matrix instructions, min trees and ballots interleave
// freely, two accumulator sets); a tile whose ballot says that some row can enter a best-two list is computed AGAIN
// on the rare path behind the block, and updated there.
template <int NQB, int WPS>
__global__ __launch_bounds__(256, WPS) void knn2_fp4s_kernel(const uint32_t* __restrict__ query, int64_t nq,
                                                             const uint32_t* __restrict__ db, int64_t ndb,
                                                             int64_t chunkRows, int64_t indexBase,
                                                             Partial* __restrict__ partial) {
    __shared__ uint32_t lut[256];
    __shared__ int pqs[4][NQB][64];
    __shared__ unsigned keys[4][NQB][2][64];       // best two keys per lane and query block: only the rare path touches them
    {
        const unsigned t = threadIdx.x;
        uint32_t v = 0;
        for (int i = 0; i < 8; i++) v |= ((t >> i) & 1u) << (4 * i + 1);
        lut[t] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t qtile = ((int64_t)blockIdx.x * 4 + wv) * (32 * NQB);
    if (qtile >= nq) return;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    uint32_t B[NQB][4][4];
    float thr[NQB];                                 // (distance of k2) - |q|, as the accumulators hold it
#pragma unroll
    for (int b = 0; b < NQB; b++) {
        const int64_t qi = min(qtile + 32 * b + j, nq - 1);
        const u4* qp = reinterpret_cast<const u4*>(query + qi * 8);
        const u4 lo = qp[0], hi = qp[1];
        const int pq = __popc(lo.x) + __popc(lo.y) + __popc(lo.z) + __popc(lo.w) + __popc(hi.x) + __popc(hi.y) + __popc(hi.z) + __popc(hi.w);
        pqs[wv][b][lane] = pq;
        const u4 mine = h ? hi : lo;
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int k = 0; k < 4; k++) B[b][m][k] = 0x22222222u | (lut[(mine[m] >> (8 * k)) & 255u] << 2);
        keys[wv][b][0][lane] = keys[wv][b][1][lane] = kEmptyKey;
        thr[b] = (float)((int)(kEmptyKey >> 23) - pq);
    }
    const int64_t j0 = (int64_t)blockIdx.y * chunkRows;
    const int rows = (int)min(chunkRows, ndb - j0);
    const int nrb = (rows + 31) >> 5;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(db + j0 * 8), 0, rows * 32, 0x00027000);
    const int laneOff = j * 32 + 16 * h;
    u4 raw = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, laneOff, 0, 0));

    auto tile = [&](const uint32_t (&A)[4][4], int b) {
        v16f acc = {0};
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const v8i a = {(int)A[m][0], (int)A[m][1], (int)A[m][2], (int)A[m][3], 0, 0, 0, 0};
            const v8i bb = {(int)B[b][m][0], (int)B[b][m][1], (int)B[b][m][2], (int)B[b][m][3], 0, 0, 0, 0};
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bb, acc, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
        return acc;
    };
    auto tile_min = [&](const v16f& acc) {
        const float m0 = fminf(fminf(acc[0], acc[1]), acc[2]), m1 = fminf(fminf(acc[3], acc[4]), acc[5]);
        const float m2 = fminf(fminf(acc[6], acc[7]), acc[8]), m3 = fminf(fminf(acc[9], acc[10]), acc[11]);
        const float m4 = fminf(fminf(acc[12], acc[13]), acc[14]);
        return fminf(fminf(fminf(m0, m1), m2), fminf(fminf(m3, m4), acc[15]));
    };

    for (int rb = 0; rb < nrb; rb++) {
        uint32_t A[4][4];
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int k = 0; k < 4; k++) A[m][k] = lut[(raw[m] >> (8 * k)) & 255u];
        if (rb + 1 < nrb) raw = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, laneOff, (rb + 1) * 1024, 0));
        // Two accumulator sets: the matrix instructions of tile b are issued, then the min tree of tile b - 1 runs
        // while they execute; nothing crosses the fences, so no third set is ever live.
        unsigned long long hit[NQB];
        v16f prev = tile(A, 0);
#pragma unroll
        for (int b = 1; b < NQB; b++) {
            __builtin_amdgcn_sched_barrier(0);
            const v16f cur = tile(A, b);
            hit[b - 1] = __builtin_amdgcn_ballot_w64(tile_min(prev) < thr[b - 1]);
            prev = cur;
        }
        __builtin_amdgcn_sched_barrier(0);
        hit[NQB - 1] = __builtin_amdgcn_ballot_w64(tile_min(prev) < thr[NQB - 1]);
        unsigned long long anyHit = 0;
#pragma unroll
        for (int b = 0; b < NQB; b++) anyHit |= hit[b];
        if (anyHit != 0) {
            const int rowBase = rb * 32 + 4 * h;
#pragma unroll
            for (int b = 0; b < NQB; b++) {
                if (hit[b] == 0) continue;
                const v16f acc = tile(A, b);
                const int pq = pqs[wv][b][lane];
                unsigned ka = keys[wv][b][0][lane], kb = keys[wv][b][1][lane];
                const unsigned base = ((unsigned)pq << 23) + (unsigned)rowBase;
                if (__popcll(hit[b]) > 6) {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int row = rowBase + 8 * (r >> 2) + (r & 3);
                        unsigned key = ((unsigned)(int)acc[r] << 23) + base + (unsigned)(8 * (r >> 2) + (r & 3));
                        if (row >= rows) key = kEmptyKey;
                        kb = max(ka, min(kb, key));     // the median of three, given ka <= kb
                        ka = min(ka, key);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        if (__builtin_amdgcn_ballot_w64(acc[r] < thr[b]) != 0) {
                            const int row = rowBase + 8 * (r >> 2) + (r & 3);
                            unsigned key = ((unsigned)(int)acc[r] << 23) + base + (unsigned)(8 * (r >> 2) + (r & 3));
                            if (row >= rows) key = kEmptyKey;
                            kb = max(ka, min(kb, key));
                            ka = min(ka, key);
                        }
                    }
                }
                keys[wv][b][0][lane] = ka;
                keys[wv][b][1][lane] = kb;
                thr[b] = (float)((int)(kb >> 23) - pq);
            }
        }
    }
#pragma unroll
    for (int b = 0; b < NQB; b++) {
        const unsigned ka = keys[wv][b][0][lane], kb = keys[wv][b][1][lane];
        const unsigned o1 = (unsigned)__shfl_xor((int)ka, 32), o2 = (unsigned)__shfl_xor((int)kb, 32);
        const unsigned n1 = min(ka, o1), n2 = min(max(ka, o1), min(kb, o2));
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

// the fold of the product (same rule), so that the probe's results can be compared end to end
__global__ void fold_kernel(const Partial* partial, int nchunks, int64_t nq, int32_t init, int64_t* idx, int32_t* d1, int32_t* d2) {
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
    idx[qi] = bi; d1[qi] = b1; d2[qi] = b2;
}

// dword d of row r -> sixteen bytes: its four bytes through the nibble table (bit -> 0b0010)
__global__ void expand_rows_kernel(const uint32_t* __restrict__ db, int64_t ndw, uint32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ndw) return;
    const uint32_t v = db[i];
    uint32_t o[4];
    for (int k = 0; k < 4; k++) {
        const uint32_t byte = (v >> (8 * k)) & 255u;
        uint32_t e = 0;
        for (int b = 0; b < 8; b++) e |= ((byte >> b) & 1u) << (4 * b + 1);
        o[k] = e;
    }
    reinterpret_cast<uint4*>(out)[i] = make_uint4(o[0], o[1], o[2], o[3]);
}

static uint64_t sm64(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int64_t nq = argc > 2 ? atoll(argv[2]) : 1000;
    const int64_t ndb = argc > 3 ? atoll(argv[3]) : 5000;
    int64_t chunkRows = argc > 4 ? atoll(argv[4]) : 1 << 22;
    const bool check = argc > 5 && atoi(argv[5]) != 0;
    const int expand = argc > 6 ? atoi(argv[6]) : 0;
    constexpr int NQB0 = 4, NQB1 = 8;
    std::vector<uint64_t> hq(nq * 4), hdb(ndb * 4);
    uint64_t s = 1;
    for (auto& v : hdb) v = sm64(s);
    for (int64_t i = 0; i < nq; i++) {
        if (i % 2 == 0 && ndb > 0) {                 // a database row with up to 40 flipped bits
            const int64_t r = (int64_t)(sm64(s) % (uint64_t)ndb);
            for (int k = 0; k < 4; k++) hq[4 * i + k] = hdb[4 * r + k];
            const int flips = (int)(sm64(s) % 41);
            for (int f = 0; f < flips; f++) { const int bit = (int)(sm64(s) % 256); hq[4 * i + bit / 64] ^= 1ull << (bit % 64); }
        } else {
            for (int k = 0; k < 4; k++) hq[4 * i + k] = sm64(s);
        }
    }
    if (check && ndb >= 8) {                          // duplicates and ties
        for (int k = 0; k < 4; k++) { hdb[4 * 5 + k] = hdb[4 * 2 + k]; hdb[4 * (ndb - 1) + k] = hdb[4 * 2 + k]; hq[k] = hdb[4 * 2 + k]; }
    }
    const int nchunks = (int)((ndb + chunkRows - 1) / chunkRows);
    uint32_t *dq, *ddb;
    Partial* part;
    int64_t* didx;
    int32_t *dd1, *dd2;
    CK(hipMalloc(&dq, nq * 32));
    CK(hipMalloc(&ddb, std::max<int64_t>(ndb * 32, 32)));
    CK(hipMalloc(&part, sizeof(Partial) * nq * std::max(nchunks, 1)));
    CK(hipMalloc(&didx, nq * 8));
    CK(hipMalloc(&dd1, nq * 4));
    CK(hipMalloc(&dd2, nq * 4));
    CK(hipMemcpy(dq, hq.data(), nq * 32, hipMemcpyHostToDevice));
    CK(hipMemcpy(ddb, hdb.data(), ndb * 32, hipMemcpyHostToDevice));
    uint32_t* ddbx = nullptr;
    if (mode == 17 || mode == 18) {
        CK(hipMalloc(&ddbx, std::max<int64_t>(ndb * 128, 128)));
        hipLaunchKernelGGL(expand_rows_kernel, dim3((unsigned)((ndb * 8 + 255) / 256)), dim3(256), 0, 0, ddb, ndb * 8, ddbx);
        CK(hipDeviceSynchronize());
    }
    auto run = [&]() {
        auto go = [&](auto kernel, int nqb) {
            dim3 grid((unsigned)((nq + 4 * 32 * nqb - 1) / (4 * 32 * nqb)), nchunks);
            hipLaunchKernelGGL(kernel, grid, dim3(256), 0, 0, dq, nq, ddb, ndb, chunkRows, 0, part);
        };
        if (mode == 17 || mode == 18) {
            const int nqb = 8;
            dim3 grid((unsigned)((nq + 4 * 32 * nqb - 1) / (4 * 32 * nqb)), nchunks);
            if (mode == 17) hipLaunchKernelGGL((knn2_fp4_kernel<8, false, 2, true, 3>), grid, dim3(256), 0, 0, dq, nq, ddbx, ndb, chunkRows, 0, part);
            else hipLaunchKernelGGL((knn2_fp4_kernel<8, true, 2, true, 3>), grid, dim3(256), 0, 0, dq, nq, ddbx, ndb, chunkRows, 0, part);
        }
        else if (mode == 2) go(knn2_fp4_kernel<8, false, 2>, 8);
        else if (mode == 3) go(knn2_fp4_kernel<6, true, 2>, 6);
        else if (mode == 4) go(knn2_fp4_kernel<2, true, 2>, 2);
        else if (mode == 5) go(knn2_fp4_kernel<1, false, 2>, 1);
        else if (mode == 6) go(knn2_fp4_kernel<16, true, 1>, 16);
        else if (mode == 7) go(knn2_fp4_kernel<4, true, 2>, 4);
        else if (mode == 8) go(knn2_fp4_kernel<8, true, 2>, 8);
        else if (mode == 9) go(knn2_fp4s_kernel<8, 2>, 8);
        else if (mode == 12) go(knn2_fp4_kernel<8, false, 2, true>, 8);
        else if (mode == 14) go(knn2_fp4_kernel<8, false, 2, true, 1>, 8);
        else if (mode == 15) go(knn2_fp4_kernel<6, true, 2, true, 2>, 6);
        else if (mode == 16) go(knn2_fp4_kernel<4, true, 2, true, 2>, 4);
        else if (mode == 13) go(knn2_fp4_kernel<2, false, 2, true>, 2);
        else if (mode == 10) go(knn2_fp4s_kernel<2, 2>, 2);
        else if (mode == 11) go(knn2_fp4s_kernel<4, 2>, 4);
        else if (mode == 0) {
            dim3 grid((unsigned)((nq + 4 * 32 * NQB0 - 1) / (4 * 32 * NQB0)), nchunks);
            hipLaunchKernelGGL((knn2_mfma_kernel<0, NQB0, 0>), grid, dim3(256), 0, 0, dq, nq, ddb, ndb, chunkRows, 0, part);
        } else {
            dim3 grid((unsigned)((nq + 4 * 32 * NQB1 - 1) / (4 * 32 * NQB1)), nchunks);
            if (expand == 0) hipLaunchKernelGGL((knn2_mfma_kernel<1, NQB1, 0>), grid, dim3(256), 0, 0, dq, nq, ddb, ndb, chunkRows, 0, part);
            else if (expand == 1) hipLaunchKernelGGL((knn2_mfma_kernel<1, NQB1, 1>), grid, dim3(256), 0, 0, dq, nq, ddb, ndb, chunkRows, 0, part);
            else hipLaunchKernelGGL((knn2_mfma_kernel<1, NQB1, 2>), grid, dim3(256), 0, 0, dq, nq, ddb, ndb, chunkRows, 0, part);
        }
        hipLaunchKernelGGL(fold_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, 0, part, nchunks, nq, 256, didx, dd1, dd2);
    };
    run();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = check ? 1 : 5;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) run();
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    long long bad = 0;
    if (check) {
        std::vector<int64_t> idx(nq);
        std::vector<int32_t> d1(nq), d2(nq);
        CK(hipMemcpy(idx.data(), didx, nq * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(d1.data(), dd1, nq * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(d2.data(), dd2, nq * 4, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < nq; i++) {
            int b1 = 256, b2 = 256;
            int64_t bi = -1;
            for (int64_t r = 0; r < ndb; r++) {
                int d = 0;
                for (int k = 0; k < 4; k++) d += __builtin_popcountll(hq[4 * i + k] ^ hdb[4 * r + k]);
                if (d < b1) { b2 = b1; b1 = d; bi = r; } else if (d < b2) b2 = d;
            }
            if (bi != idx[i] || b1 != d1[i] || b2 != d2[i]) {
                if (bad < 5) fprintf(stderr, "query %lld: got (%lld, %d, %d) want (%lld, %d, %d)\n", (long long)i, (long long)idx[i], d1[i], d2[i], (long long)bi, b1, b2);
                bad++;
            }
        }
    }
    printf("{\"mode\": \"%s\", \"expand\": %d, \"nq\": %lld, \"ndb\": %lld, \"chunk_rows\": %lld, \"ms\": %.3f, \"pairs_per_s\": %.4g, \"checked\": %s, \"wrong\": %lld}\n",
           mode == 0 ? "i8" : mode == 1 ? "fp4" : mode == 2 ? "fp4_v2_8" : mode == 3 ? "fp4_v2_6pair" : mode == 4 ? "fp4_v2_2pair" : mode == 5 ? "fp4_v2_1" : mode == 6 ? "fp4_v2_16pair_1wave" : mode == 7 ? "fp4_v2_4pair" : mode == 8 ? "fp4_v2_8pair" : mode == 9 ? "fp4_v3_8" : mode == 10 ? "fp4_v3_2" : mode == 11 ? "fp4_v3_4" : mode == 12 ? "fp4_v2_8_unscaled" : mode == 13 ? "fp4_v2_2_unscaled" : mode == 14 ? "fp4_v2_8_no_update_path" : mode == 15 ? "fp4_v2_6_joint_pair" : mode == 17 ? "fp4_v2_8_rows_kept_expanded" : mode == 18 ? "fp4_v2_8pair_rows_kept_expanded" : "fp4_v2_4_joint_pair", expand, (long long)nq, (long long)ndb, (long long)chunkRows, ms, (double)nq * ndb / (ms * 1e-3),
           check ? "true" : "false", bad);
    return bad ? 3 : 0;
}
